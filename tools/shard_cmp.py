"""Kernel time of every tile shard of the headline frame (Cornell 1024x1024 @1000) against the whole frame, for
splits over 2, 4 and 8, and of the matching sample slices: what a rank of a multi-GPU run has to render, measured
on one GPU. The slowest shard sets the step time of the run. Usage: PYTHONPATH=. python tools/shard_cmp.py"""
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

ctx = lib.Context(0)
cfg, flat = setup("cornell-box", 1)
cam = build_camera(cfg, 1.0)
sc = ctx.scene(flat)


def run(**kw):
    p = abi.make_params(1024, 1024, kw.pop("spp"), 50, cfg["background"], seed=0x5EED0001, **kw)
    sc.render_device(cam, p, 0, 0, want_stats=True)
    st = [sc.render_device(cam, p, 0, 0, want_stats=True) for _ in range(3)]
    mhz.append(min(s["shader_mhz"] for s in st))
    return min(s["kernel_ms"] for s in st)


mhz = []  # the shader clock each launch measured in itself (lowest of the three), in the order of the runs


full = run(spp=1000)
print(f"whole frame @1000: kernel {full:.3f} ms", flush=True)
for n in (2, 4, 8):
    ms = [run(spp=1000, shard_index=r, shard_count=n) for r in range(n)]
    print(f"tile split over {n}: shards " + " ".join(f"{m:.3f}" for m in ms) +
          f" ms; slowest {max(ms):.3f} = {full / n / max(ms):.3f} of the whole frame's rate", flush=True)
    s = run(spp=1000 // n)
    print(f"sample slice of {1000 // n} spp: {s:.3f} ms = {full / n / s:.3f}", flush=True)
print("shader MHz per run: " + " ".join(f"{m:.0f}" for m in mhz))
