"""The three tree builders side by side on the worlds with trees: kernel time of a render, box tests per ray
(event counters), tree height, and that the frames are bit-identical.
Usage: python tools/builder_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

CASES = {"random-scene": (1200, 800, 64), "final-scene": (1920, 1080, 32)}
ctx = lib.Context(0)
for name, (W, H, spp) in CASES.items():
    cfg, flat = setup(name, 1)
    cam = build_camera(cfg, W / H)
    frames = {}
    for label, b in (("host SAH", abi.RT_BVH_HOST_SAH), ("device SAH", abi.RT_BVH_DEVICE_SAH), ("device LBVH", abi.RT_BVH_DEVICE_LBVH)):
        t0 = time.perf_counter()
        sc = ctx.scene(flat, bvh_builder=b)
        t_scene = (time.perf_counter() - t0) * 1e3
        p = abi.make_params(W, H, spp, 50, cfg["background"], seed=5)
        sc.render(cam, abi.make_params(W, H, 1, 50, cfg["background"]), want_argb=False)
        best = min(sc.render_device(cam, p, 0, 0, want_stats=True)["kernel_ms"] for _ in range(3))
        small = abi.make_params(160, 120, 4, 50, cfg["background"], seed=5)
        rgb, _, _ = sc.render(cam, small)
        cst = sc.render_device(cam, abi.make_params(W // 4, H // 4, 4, 50, cfg["background"], seed=5, flags=abi.RT_FLAG_COUNTERS), 0, 0, want_stats=True)
        frames[label] = rgb
        print(f"{name:14s} {label:11s}: kernel {best:8.2f} ms  {W * H * spp / best / 1e3:8.1f} Msamples/s  box tests/ray {cst['events']['aabb_test'] / cst['rays']:6.2f}  "
              f"LDS {sc.info()['lds_bytes']}  tree {sc.bvh_digest()}  scene create {t_scene:.1f} ms", flush=True)
        sc.close()
    same = all(np.array_equal(frames['host SAH'].view(np.uint32), frames[k].view(np.uint32)) for k in frames)
    print(f"{name:14s} frames bit-identical: {same}")
ctx.close()
