"""Kernel time of one shard of N of the Cornell headline job for growing N: what a launch costs beyond
its share of the work (start-up and tail), about 0.6 ms whatever the share."""
import sys
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup
cfg, flat = setup("cornell-box", 1)
cam = build_camera(cfg, 1.0)
ctx = lib.Context(0)
sc = ctx.scene(flat)
W = H = 1024; spp = 1000
def t(**kw):
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=0x5EED0001, **kw)
    sc.render_device(cam, p, 0, 0, want_stats=True)
    return min(sc.render_device(cam, p, 0, 0, want_stats=True)["kernel_ms"] for _ in range(3))
full = t()
for n in (8, 16, 32, 64, 128):
    s = t(shard_index=1, shard_count=n)
    print(f"full {full:.2f} ms; shard 1/{n}: {s:.3f} ms; ideal {full/n:.3f}; extra {s-full/n:.3f} ms", flush=True)
