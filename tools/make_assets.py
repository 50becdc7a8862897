"""Decodes the reference's image-texture assets once, in the build container, into raw RGB8
arrays shipped with the package (the GPU box has neither the reference tree nor a need for a
JPEG decoder). Pillow's decoder stands in for javax.imageio (reference texture.clj:72); the two
may differ by one LSB per texel.

Usage: python tools/make_assets.py [/root/reference/resources]
"""
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/resources"
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raytracing_amd", "assets")
os.makedirs(dst, exist_ok=True)
for name in ("earthmap.jpg", "stones.jpg", "map2.png"):
    with Image.open(os.path.join(src, name)) as im:
        rgb = np.ascontiguousarray(np.asarray(im.convert("RGB")), dtype=np.uint8)
    out = os.path.join(dst, os.path.splitext(name)[0] + "_rgb8.npz")
    np.savez_compressed(out, rgb8=rgb)
    print(name, rgb.shape, "->", out, os.path.getsize(out), "bytes")
