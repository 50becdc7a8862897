"""Throughput of the path kernel on every reference world at its full benchmark size
(reduced spp): Msamples/s, rays/sample, kernel ms. Usage: python tools/scene_bench.py [scene ...]"""
import sys
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

CASES = {  # scene -> (W, H, spp)
    "cornell-box": (1024, 1024, 64),
    "cornell-smoke": (1024, 1024, 32),
    "random-scene": (1200, 800, 32),
    "two-spheres": (1024, 1024, 32),
    "two-perlin-spheres": (1024, 1024, 16),
    "earth": (1024, 1024, 32),
    "simple-light": (1024, 1024, 16),
    "final-scene": (3840, 2160, 4),
}
names = sys.argv[1:] or list(CASES)
ctx = lib.Context(0)
for name in names:
    W, H, spp = CASES[name]
    cfg, flat = setup(name, 1)
    cam = build_camera(cfg, W / H)
    sc = ctx.scene(flat)
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001)
    sc.render_device(cam, p, 0, 0, want_stats=True)  # warm-up
    st = sc.render_device(cam, p, 0, 0, want_stats=True)
    ms = st["kernel_ms"]
    print(f"{name:20s} {W}x{H} @{spp:3d}: kernel {ms:9.2f} ms  {W * H * spp / ms / 1e3:9.1f} Msamples/s  "
          f"{st['rays'] / ms / 1e3:9.1f} Mrays/s  rays/sample {st['rays'] / st['samples']:6.2f}  info {sc.info()}", flush=True)
    sc.close()
ctx.close()
