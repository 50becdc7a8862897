"""Kernel time of the Cornell headline job with the library named by RTMI_LIB."""
import os
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup
cfg, flat = setup("cornell-box", 1)
cam = build_camera(cfg, 1.0)
ctx = lib.Context(0)
sc = ctx.scene(flat)
p = abi.make_params(1024, 1024, 1000, 50, cfg["background"], seed=0x5EED0001)
sc.render_device(cam, p, 0, 0, want_stats=True)
ms = min(sc.render_device(cam, p, 0, 0, want_stats=True)["kernel_ms"] for _ in range(2))
print(os.environ.get("RTMI_LIB", "default"), f"{ms:.2f} ms", flush=True)
