"""Throughput of the path kernel on every reference world at its full benchmark size
(reduced spp): Msamples/s, rays/sample, kernel ms. Usage: python tools/scene_bench.py [--full] [--pool] [scene ...]
--full: the sample counts of the benchmark configurations (random-scene @500, the texture worlds and the
Cornell boxes @1000; final-scene @100 of its 10 000, one GPU's share of a sample split being 1 250).
--spp=N: that sample count for every scene named."""
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
FULL = {"cornell-box": 1000, "cornell-smoke": 1000, "random-scene": 500, "two-spheres": 1000,
        "two-perlin-spheres": 1000, "earth": 1000, "simple-light": 1000, "final-scene": 100}
full = "--full" in sys.argv[1:]
pool = "--pool" in sys.argv[1:]   # the re-packing kernel (RT_FLAG_POOL_PATHS) where the world can use it
spp_arg = [int(a.split("=", 1)[1]) for a in sys.argv[1:] if a.startswith("--spp=")]
names = [a for a in sys.argv[1:] if a not in ("--full", "--pool") and not a.startswith("--spp=")] or list(CASES)
ctx = lib.Context(0)
for name in names:
    W, H, spp = CASES[name]
    if full:
        spp = FULL[name]
    if spp_arg:
        spp = spp_arg[0]
    cfg, flat = setup(name, 1)
    cam = build_camera(cfg, W / H)
    sc = ctx.scene(flat)
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001, flags=abi.RT_FLAG_POOL_PATHS if pool else 0)
    sc.render_device(cam, p, 0, 0, want_stats=True)  # warm-up
    st = sc.render_device(cam, p, 0, 0, want_stats=True)
    ms = st["kernel_ms"]
    print(f"{name:20s} {W}x{H} @{spp:4d}: kernel {ms:9.2f} ms  {W * H * spp / ms / 1e3:9.1f} Msamples/s  "
          f"{st['rays'] / ms / 1e3:9.1f} Mrays/s  rays/sample {st['rays'] / st['samples']:6.2f}  info {sc.info()}", flush=True)
    sc.close()
ctx.close()
