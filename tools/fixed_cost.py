"""Kernel time of the Cornell frame (1024x1024) against samples per pixel and depth limit: what a launch costs beyond
its samples, and how the cost of a sample depends on the size of the work items (which follows from spp).
Usage: PYTHONPATH=. python tools/fixed_cost.py"""
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup
ctx = lib.Context(0)
cfg, flat = setup("cornell-box", 1); cam = build_camera(cfg, 1.0); sc = ctx.scene(flat)
for depth in (50, 8, 1):
    for spp in (1, 2, 4, 8, 16, 32, 64, 125, 250):
        p = abi.make_params(1024, 1024, spp, depth, cfg["background"], seed=0x5EED0001)
        sc.render_device(cam, p, 0, 0, want_stats=True)
        st = [sc.render_device(cam, p, 0, 0, want_stats=True) for _ in range(3)]
        ms = min(s["kernel_ms"] for s in st)
        print(f"depth {depth:2d} spp {spp:4d}: kernel {ms:8.3f} ms  per spp {ms/spp:7.4f}  rays/sample {st[0]['rays']/st[0]['samples']:.2f}", flush=True)
