"""The re-packing kernel against the lane-owns-a-path kernel: same frame bit for bit, and the speed of both.
Usage: python tools/pool_check.py [scene ...]"""
import sys
import time
import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch  # noqa: F401

from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

SIZES = {"random-scene": (1200, 800, 64), "final-scene": (1920, 1080, 32)}
ctx = lib.Context(0)
for name in (sys.argv[1:] or ["random-scene", "final-scene"]):
    cfg, flat = setup(name, 1)
    W, H, spp = SIZES.get(name, (512, 512, 32))
    cam = build_camera(cfg, W / H)
    sc = ctx.scene(flat)
    small = {}
    for flags in (0, abi.RT_FLAG_POOL_PATHS):
        p = abi.make_params(61, 45, 7, 50, cfg["background"], seed=5, flags=flags)
        rgb, argb, st = sc.render(cam, p)
        small[flags] = (rgb, st)
        print(name, "flags", flags, "small: samples", st["samples"], "rays", st["rays"], "ms", round(st["kernel_ms"], 3), flush=True)
    same = np.array_equal(small[0][0].view(np.uint32), small[2][0].view(np.uint32))
    print(name, "bit-identical:", same, "rays equal:", small[0][1]["rays"] == small[2][1]["rays"], flush=True)
    if not same:
        d = np.abs(small[0][0] - small[2][0]).max(axis=2)
        print("  pixels differing:", int((d > 0).sum()), "max", d.max())
    for flags in (0, abi.RT_FLAG_POOL_PATHS):
        p = abi.make_params(W, H, spp, 50, cfg["background"], seed=5, flags=flags)
        sc.render(cam, abi.make_params(W, H, 1, 50, cfg["background"], flags=flags), want_argb=False)
        best = 1e9
        for _ in range(3):
            _, _, st = sc.render(cam, p, want_argb=False)
            best = min(best, st["kernel_ms"])
        print(f"{name} {W}x{H}@{spp} {'pool      ' if flags else 'lane-paths'}: kernel {best:8.2f} ms  {W * H * spp / best / 1e3:9.1f} Msamples/s", flush=True)
    sc.close()
ctx.close()
