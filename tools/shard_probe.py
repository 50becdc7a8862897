"""Kernel time of the whole Cornell headline job and of one shard of 8, to size the work items."""
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
    return min(sc.render_device(cam, p, 0, 0, want_stats=True)["kernel_ms"] for _ in range(2))
full = t()
for n in (2, 4, 8):
    s = t(shard_index=1, shard_count=n)
    print(f"full {full:.2f} ms; shard 1/{n}: {s:.2f} ms; efficiency {full / n / s:.3f}", flush=True)
