"""Wall time of one rt_render_device call against its path-kernel time, for one shard of N of the
headline job: what a rank's step costs beyond the kernel (launches, resolve, stats read-back)."""
import time
import torch
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

cfg, flat = setup("cornell-box", 1)
cam = build_camera(cfg, 1.0)
ctx = lib.Context(0)
sc = ctx.scene(flat)
W = H = 1024
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
argb = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for n in (1, 2, 4, 8):
    p = abi.make_params(W, H, 1000, 50, cfg["background"], seed=0x5EED0001, shard_index=0, shard_count=n)
    sc.render_device(cam, p, acc.data_ptr(), 0, want_stats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ks = []
    for _ in range(5):
        st = sc.render_device(cam, p, acc.data_ptr(), 0, want_stats=True)
        ctx.tonemap_device(acc.data_ptr(), W * H, 1000, argb.data_ptr(), 0)
        ks.append(st["kernel_ms"])
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    k = sum(ks) / len(ks)
    print(f"shard 1/{n}: step {wall:.3f} ms, path kernel {k:.3f} ms, rest {wall - k:.3f} ms; ideal share {ks[0] if n == 1 else 0:.1f}", flush=True)
