#!/usr/bin/env python3
"""Headline benchmark: Cornell box 1024x1024 @ 1000 spp (the configuration the headline metric is quoted on), Msamples/s.

A step = one complete render of that workload: every rank renders its tile shard (rank r owns
one 8x8 tile of every N consecutive ones, the place moving on by one per tile row), the per-GPU
FP32 accumulators are summed onto rank 0 with one
RCCL reduce and rank 0 packs the pixels. Inputs (the uploaded world, the camera) are resident in
HBM before the timed region. Total work is fixed as N grows (strong scaling).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (before librtmi.so: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

from raytracing_amd import abi, lib  # noqa: E402
from raytracing_amd.core import build_camera, setup  # noqa: E402
from raytracing_amd.dist import render_distributed, shard_params as dist_shard_params  # noqa: E402

# Algorithmic FLOPs per event of the path kernel (add/sub/mul/div/sqrt/min/max/abs/floor and each
# transcendental = 1, FMA = 2; compares, selects, integer and RNG work = 0), following the
# reference's arithmetic with common subexpressions computed once.
FLOPS_PER_EVENT = {
    "aabb_test": 21, "sphere_test": 23, "sphere_root": 4, "sphere_record": 32, "msphere_center": 12,
    "rect_test": 6, "rect_record": 20, "rotate_y_ray": 12, "rotate_y_rec": 20, "translate_ray": 3,
    "translate_rec": 11, "medium_sample": 16, "camera_ray": 35, "unit_sphere_trial": 11,
    "unit_disk_trial": 7, "lambert_scatter": 13, "metal_scatter": 33, "dielectric_scatter": 60,
    "iso_scatter": 0, "light_emit": 0, "checker": 8, "perlin_noise": 250, "noise_tex": 41,
    "image_tex": 10, "throughput_mul": 3, "accumulate": 3,
    # six rects that close a box as ONE slab test (what the kernel executes): 6 sub + 6 mul for the
    # plane parameters, 6 min/max per axis pair, 2 + 2 for entry and exit, 3 sub in the range test
    "box_slab": 25,
}


def fp32_peak_tflops(dev):
    """Vector FP32 peak of the device in use: CUs x 4 SIMDs x 64 FLOP/clk (16 lanes x FMA x packed
    v_pk_fma_f32) x the clock the device reports. MI355X: 256 x 4 x 64 x 2.4 GHz = 157.3."""
    return dev["n_cu"] * 4 * 64 * dev["clock_khz"] * 1e3 / 1e12


# The other single-GPU benchmark configurations (kernel time of one render each, same run).
OTHER_CONFIGS = [
    ("C1", "final-scene", 200, 200, 10, "tile"),          # (-main :samples 10), core.clj:50-59
    ("C2", "random-scene", 1200, 800, 500, "tile"),
    ("C3", "two-perlin-spheres", 1024, 1024, 1000, "tile"),
    ("C3", "earth", 1024, 1024, 1000, "tile"),
    ("C3", "simple-light", 1024, 1024, 1000, "tile"),
    ("C5/1gpu", "final-scene", 3840, 2160, 100, "sample"),  # one GPU's view of the 4K frame, 100 of the 10 000 spp
]


def pmc_record(scene, W, H):
    """Counters of the profiled launch of this workload (profiles/pmc.json, written by
    tools/make_pmc_json.py from the rocprofv3 --pmc passes), or {}."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc.json"))).get(f"{scene} {W}x{H}", {})
    except Exception:
        return {}


def measure_config(ctx, device, scene_name, W, H, spp, peak):
    """One render of another configuration on this GPU: kernel rate, FLOP by event count, roofline."""
    cfg, flat = setup(scene_name, scene_seed=1)
    cam = build_camera(cfg, W / H)
    sc = ctx.scene(flat)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001)
    sc.render_device(cam, abi.make_params(W, H, min(spp, 2), 50, cfg["background"]), acc.data_ptr(), 0, want_stats=True, stream=stream)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    st = sc.render_device(cam, p, acc.data_ptr(), 0, want_stats=True, stream=stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    cp = abi.make_params(W, H, min(spp, 8), 50, cfg["background"], seed=0x5EED0001, flags=abi.RT_FLAG_COUNTERS)
    cst = sc.render_device(cam, cp, 0, 0, want_stats=True, stream=stream)
    fps = sum(FLOPS_PER_EVENT[k] * v for k, v in cst["events"].items()) / max(cst["samples"], 1)
    rv, _ = sc.variants()
    sc.close()
    del acc
    samples = W * H * spp
    achieved = fps * samples / (st["kernel_ms"] * 1e-3) / 1e12
    pm = pmc_record(scene_name, W, H)
    rec = {"workload": f"{scene_name} {W}x{H} @{spp}spp depth 50", "value": samples / wall / 1e6, "unit": "Msamples/s",
           "kernel_ms": st["kernel_ms"], "kernel_Msamples_per_s": samples / (st["kernel_ms"] * 1e-3) / 1e6,
           "rays_per_sample": st["rays"] / samples, "flop_per_sample": fps, "achieved_tflops": achieved,
           "frac": achieved / peak, "frac_nonpacked": achieved / (peak / 2),
           "kernel": f"rt::k_render<{abi.VARIANT_NAMES[rv]}, false>", "shader_mhz": st["shader_mhz"]}
    if pm.get("active_lane_ratio") is not None:
        # not measured by this run: a figure of an earlier rocprofv3 --pmc pass, quoted with where it comes from
        rec["from_profile"] = {"active_lane_ratio": pm["active_lane_ratio"], "profiled_spp": pm.get("profiled_spp"),
                               "file": "profiles/pmc.json", "source": pm.get("source"), "commit": pm.get("commit")}
    if st["kernel_ms"] < 2.0:
        rec["launch_bound"] = (f"{samples} samples are {samples / (256 * 4 * 64 * 4):.1f} per resident lane of a launch that "
                               "fills the GPU: the figure prices a launch, not the kernel's rate")
    return rec


def cpu_baseline(flat, cam, cfg, seconds=12.0):
    """The C++ FP64 restatement of the reference path (oracle/), timed on this host's cores on a
    bounded sample of the same workload: Cornell at 256x256, spp chosen for ~`seconds` of work."""
    from oracle import binding as ob
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    threads = max(1, min(threads, os.cpu_count() or 1))
    W = H = 256
    pilot = abi.make_params(64, 64, 8, 50, cfg["background"], seed=1)
    t0 = time.perf_counter()
    ob.render(flat, cam, pilot, nthreads=threads, want_argb=False)
    rate = 64 * 64 * 8 / max(time.perf_counter() - t0, 1e-6)
    spp = int(max(4, min(1000, seconds * rate / (W * H))))
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=1)
    t0 = time.perf_counter()
    _, _, rays = ob.render(flat, cam, p, nthreads=threads, want_argb=False)
    dt = time.perf_counter() - t0
    out = {"value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": f"cornell-box {W}x{H} @{spp}spp depth 50 ({W * H * spp} samples, {dt:.1f} s), "
                     "C++ FP64 restatement of the Clojure path, not the JVM",
           "rays_per_sample": rays / (W * H * spp)}
    # the reference's own runnable case, (-main :samples 10) = final-scene 200x200 @10 (core.clj:50-59)
    cfg1, flat1 = setup("final-scene", scene_seed=1)
    p1 = abi.make_params(200, 200, 10, 50, cfg1["background"], seed=0x5EED0001)
    t0 = time.perf_counter()
    ob.render(flat1, build_camera(cfg1, 1.0), p1, nthreads=threads, want_argb=True)
    dt1 = time.perf_counter() - t0
    out["main_samples_10"] = {"value": 200 * 200 * 10 / dt1 / 1e6, "unit": "Msamples/s", "seconds": dt1, "cores": threads,
                              "sample": "final-scene 200x200 @10spp (the whole job), C++ FP64 restatement; JVM unavailable "
                                        "on this machine, so the Clojure path itself was not timed"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="cornell-box")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--split", default="tile", choices=["tile", "sample"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-other-configs", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched through torch.distributed.run")
        args.gpus = world
    # Rehearsal hooks (a one-GPU box cannot run RCCL between two ranks on the same device):
    # RTMI_ONE_GPU=1 puts every rank on device 0, RTMI_DIST_BACKEND=gloo swaps the collective backend.
    backend = os.environ.get("RTMI_DIST_BACKEND", "nccl")
    if os.environ.get("RTMI_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg, flat = setup(args.scene, scene_seed=1)
    cam = build_camera(cfg, args.width / args.height)
    ctx = lib.Context(local_rank)
    scene = ctx.scene(flat)
    W, H, spp = args.width, args.height, args.spp
    base = abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
    argb = torch.zeros((H, W), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device)
    kernel_ms, rays_total, shader_mhz_seen = [], [], []

    def render_fn(p):
        st = scene.render_device(cam, p, acc.data_ptr(), 0, want_stats=True, stream=stream.cuda_stream)
        kernel_ms.append(st["kernel_ms"])
        rays_total.append(st["rays"])
        shader_mhz_seen.append(st["shader_mhz"])
        return acc

    # gloo has no reduce() for device tensors; all_reduce gives rank 0 the same sum
    reduce_fn = None if backend == "nccl" else (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))

    reduce_events = []  # (start, end) events of each step's reduce on this rank's stream

    def timed_reduce(t):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        if reduce_fn is None:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        else:
            reduce_fn(t)
        e1.record(stream)
        reduce_events.append((e0, e1))

    def step():
        render_distributed(render_fn, base, rank, world, args.split, timed_reduce)
        if rank == 0:
            ctx.tonemap_device(acc.data_ptr(), W * H, spp, argb.data_ptr(), stream.cuda_stream)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms.clear()
    rays_total.clear()
    shader_mhz_seen.clear()
    reduce_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(sum(kernel_ms)) / max(len(kernel_ms), 1), float(sum(rays_total)),
                      float(sum(shader_mhz_seen)) / max(len(shader_mhz_seen), 1)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, kern_ms_max, rays = float(tmax[0]), float(tmax[1]), float(tsum[2])
    else:
        kern_ms_max, rays = float(t[1]), float(t[2])

    # Per rank: the path kernel's time and the reduce's, apart (the reduce includes the wait for the slowest rank).
    # Then one more, untimed, step that checks the exchange: the reduced frame on rank 0 must add up to the sum of
    # what the ranks rendered (exactly for the tile split, whose shards own disjoint pixels).
    per_rank = None
    reduce_check = None
    if world > 1:
        my_reduce_ms = sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1)
        gathered = torch.zeros((world, 2), dtype=torch.float64, device=device)  # (all_reduce: the one collective every backend has here)
        gathered[rank, 0] = float(sum(kernel_ms)) / max(len(kernel_ms), 1)
        gathered[rank, 1] = my_reduce_ms
        dist.all_reduce(gathered, op=dist.ReduceOp.SUM)
        per_rank = [{"rank": r, "kernel_ms": float(gathered[r, 0]), "reduce_ms": float(gathered[r, 1])} for r in range(world)]
        shard = render_fn(dist_shard_params(base, rank, world, args.split))
        check = shard.double().sum().reshape(1)
        dist.all_reduce(check, op=dist.ReduceOp.SUM)
        if reduce_fn is None:
            dist.reduce(shard, dst=0, op=dist.ReduceOp.SUM)
        else:
            reduce_fn(shard)
        torch.cuda.synchronize(device)
        if rank == 0:
            got, want = float(shard.double().sum()), float(check[0])
            tol = (1e-9 if args.split == "tile" else 1e-5) * max(abs(want), 1.0)
            # (a mismatch fails the run AFTER the JSON line is out: the line then says "ok": false and the exit code says so too)
            reduce_check = {"sum_of_reduced_frame": got, "sum_of_shard_sums": want, "tolerance": tol,
                            "ok": bool(abs(got - want) <= tol)}

    # Roofline numerator: event counts of the SAME kernel from its counting build (a separate,
    # untimed launch at reduced spp; FLOP per sample does not depend on spp beyond MC noise).
    count_spp = min(spp, 16)
    cp = abi.make_params(W, H, count_spp, 50, cfg["background"], seed=0x5EED0001, shard_index=rank,
                         shard_count=world, flags=abi.RT_FLAG_COUNTERS)
    cst = scene.render_device(cam, cp, 0, 0, want_stats=True, stream=stream.cuda_stream)
    flop = sum(FLOPS_PER_EVENT[k] * v for k, v in cst["events"].items())
    flop_per_sample = flop / max(cst["samples"], 1)
    fl = torch.tensor([flop, float(cst["samples"])], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(fl, op=dist.ReduceOp.SUM)
        flop_per_sample = float(fl[0]) / max(float(fl[1]), 1.0)

    if rank == 0:
        samples_per_step = W * H * spp
        value = samples_per_step * args.steps / elapsed / 1e6
        # per-GPU kernel rate: this rank's share of the samples over its own kernel time
        my_samples = samples_per_step / world
        achieved = flop_per_sample * my_samples / (kern_ms_max * 1e-3) / 1e12
        dev = ctx.device_info()
        peak = fp32_peak_tflops(dev)
        shader_mhz = float(t[3]) if world == 1 else None
        traffic = None
        valu_cycles_per_inst = None
        active_lane_ratio = None
        from_profile = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        # the profiled launch is the headline workload: its counters say nothing about any other
        headline = (args.scene, W, H, spp) == ("cornell-box", 1024, 1024, 1000)
        if headline and os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                traffic = prof.get("k_render_hbm_bytes_per_launch")
                # SIMD cycles available per VALU wave-instruction of the profiled launch (same workload
                # at N = 1): the measured issue cost is 2.7 cycles for FMA-class and 4.5 for
                # compare/select/min/max/convert/FP64/SGPR-operand instructions, so a value near 3.3
                # means the VALU issue slots are full
                active_lane_ratio = prof.get("active_lane_ratio")
                if world == 1 and prof.get("valu_wave_insts_per_launch"):
                    # real cycles: the shader clock one wave measured during this very kernel
                    clk = (shader_mhz or dev["clock_khz"] * 1e-3) * 1e6
                    simd_cycles = kern_ms_max * 1e-3 * clk * dev["n_cu"] * 4
                    valu_cycles_per_inst = simd_cycles / prof["valu_wave_insts_per_launch"]
                if traffic is not None and world > 1:
                    traffic = traffic / world  # measured at N = 1; a rank writes its own tiles' slots only
                # what this line quotes from that file instead of measuring it: rocprofv3 --pmc cannot run inside bench.py
                from_profile = {"file": "profiles/traffic.json", "source": prof.get("source"), "commit": prof.get("commit"),
                                "kernel": prof.get("kernel"),
                                "fields": ["roofline.traffic", "roofline.active_lane_ratio",
                                           "roofline.simd_cycles_per_valu_instruction (its instruction count; the cycles are this run's)"]}
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s (whole node) Cornell 1024^2 @1000spp; % FP32 peak" if headline
                      else f"Msamples/s (whole node) {args.scene} {W}x{H} @{spp}spp; % FP32 peak",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H} @{spp}spp depth 50 (core.clj:52-59 constants, "
                                   f"reference {args.scene} world)",
                       "parallelism": f"{args.split}-shard x{world}" + (f" + {'RCCL' if backend == 'nccl' else backend} reduce(sum) to rank 0" if world > 1 else ""),
                       "samples_per_step": samples_per_step},
            "rays_per_sample": rays / (samples_per_step * args.steps),
            "Mrays_per_s": rays / elapsed / 1e6,
            "flop_per_sample": flop_per_sample,
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "frac_nonpacked": achieved / (peak / 2),
                         "peak_nonpacked": peak / 2,
                         "peak_from": f"{dev['n_cu']} CUs x 4 SIMDs x 64 FLOP/clk x {dev['clock_khz'] / 1e6:.3f} GHz "
                                      "(device properties); non-packed = half (one v_fma_f32 per lane)",
                         "shader_mhz_measured": shader_mhz, "active_lane_ratio": active_lane_ratio,
                         "traffic": traffic,
                         "kernel": f"rt::k_render<{abi.VARIANT_NAMES[scene.variants()[0]]}, false>", "kernel_ms": kern_ms_max,
                         "from_profile": from_profile,
                         "simd_cycles_per_valu_instruction": valu_cycles_per_inst,
                         "note": "algorithmic FLOP per event x event counts of the counting build of the same "
                                 "kernel, per GPU, over that GPU's HIP-event kernel time; the path is branchy "
                                 "FP32 VALU work, neither HBM- nor MFMA-bound"},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["reduce_check"] = reduce_check
        if world == 1 and headline and not args.no_other_configs:
            scene.close()   # the other worlds take their own slabs (11 GB for the 4K frame)
            out["other_configs"] = []
            for tag, name, w2, h2, spp2, split in OTHER_CONFIGS:
                rec = measure_config(ctx, device, name, w2, h2, spp2, peak)
                rec["config"] = tag
                rec["split"] = split
                out["other_configs"].append(rec)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, cam, cfg, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    scene.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if reduce_check is not None and not reduce_check["ok"]:
        sys.exit(f"reduced frame sums to {reduce_check['sum_of_reduced_frame']!r}, the ranks' shards to "
                 f"{reduce_check['sum_of_shard_sums']!r}")


if __name__ == "__main__":
    main()
