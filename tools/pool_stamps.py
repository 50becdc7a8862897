"""Where a wave of the re-packing kernel spends its cycles (diagnostic build: make -C raytracing_amd/csrc poolstamps).
Usage: RTMI_LIB=raytracing_amd/librtmi_poolstamps.so python tools/pool_stamps.py [scene]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup
SIZES = {"random-scene": (1200, 800, 32), "final-scene": (1920, 1080, 16)}
ctx = lib.Context(0)
for name in (sys.argv[1:] or ["random-scene", "final-scene"]):
    cfg, flat = setup(name, 1)
    W, H, spp = SIZES[name]
    cam = build_camera(cfg, W / H)
    sc = ctx.scene(flat)
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=5, flags=abi.RT_FLAG_POOL_PATHS)
    sc.render(cam, p, want_argb=False)
    _, _, st = sc.render(cam, p, want_argb=False)
    ev = list(st["events"].values())
    names = ["GEN", "NEW", "WALK", "LEAF", "POST"]
    total = sum(ev[0:5]) + ev[15]
    print(f"{name} {W}x{H}@{spp}: kernel {st['kernel_ms']:.2f} ms, rays/sample {st['rays'] / st['samples']:.2f}")
    for i, nm in enumerate(names):
        r = max(ev[5 + i], 1)
        print(f"  {nm:5s} {100 * ev[i] / total:5.1f} % of wave cycles, {ev[5 + i]:9d} rounds, {ev[i] / r:9.0f} cycles/round, {ev[10 + i] / r:5.1f} paths/round")
    print(f"  census+pack {100 * ev[15] / total:5.1f} %; WALK iterations {ev[16]}, lanes walking per iteration {ev[17] / max(ev[16], 1):.1f}, steps per ray {ev[17] / max(st['rays'], 1):.1f}")
    sc.close()
