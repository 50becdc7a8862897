"""Host-side mirror of the reference's entry point (src/raytracing/core.clj:50-129).

`main(samples=n)` keeps the meaning of `(-main :samples n)`: the constants of core.clj:52-59
(200x200, aspect 1.0, depth 50, :final-scene) are the defaults, and the sample loop of
core.clj:96-127 runs on the GPU through librtmi.so. The other keywords are what the reference
reaches only by editing its source.
"""
from __future__ import annotations

import random
from typing import Optional, Sequence

import numpy as np

from . import abi, camera as cam_mod, flatten, lib, scenes


def build_camera(cfg: dict, aspect_ratio: float) -> abi.rt_camera:
    """core.clj:89-91: vup [0 1 0], dist-to-focus 10.0, shutter 0.0..1.0."""
    c = cam_mod.camera(cfg["lookfrom"], cfg["lookat"], (0.0, 1.0, 0.0), cfg["vfov"], aspect_ratio,
                       cfg["aperture"], 10.0, 0.0, 1.0)
    return abi.make_camera(c)


def setup(scene: str = "final-scene", scene_seed: int = 1):
    """-> (config map, flat scene): the world selection of core.clj:60-88."""
    cfg = scenes.scene_config(scene)
    world = cfg["world"](random.Random(scene_seed))
    return cfg, flatten.flatten_world(world)


def to_buffered_image(argb: np.ndarray) -> np.ndarray:
    """The pixel placement of core.clj:92-94,123-126: a (H+75) x (W+20) TYPE_INT_RGB image,
    row y of the render written to image row (H+30)-y, starting at column 0."""
    H, W = argb.shape
    img = np.zeros((H + 75, W + 20), dtype=np.uint32)
    for y in range(H):
        img[(H + 30) - y, :W] = argb[y] & 0x00FFFFFF
    return img


def write_ppm(path: str, argb: np.ndarray) -> None:
    """Top row first (y = H-1 is the top image row; core.clj:123-126 flips the same way)."""
    H, W = argb.shape
    flipped = argb[::-1]
    rgb = np.stack([(flipped >> 16) & 255, (flipped >> 8) & 255, flipped & 255], axis=-1).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(f"P6\n{W} {H}\n255\n".encode())
        f.write(rgb.tobytes())


def write_png(path: str, argb: np.ndarray) -> None:
    """The same frame as an 8-bit RGB PNG (one zlib stream, filter 0 on every row), top row first."""
    import struct
    import zlib
    H, W = argb.shape
    flipped = argb[::-1]
    rgb = np.stack([(flipped >> 16) & 255, (flipped >> 8) & 255, flipped & 255], axis=-1).astype(np.uint8)
    raw = np.concatenate([np.zeros((H, 1), np.uint8), rgb.reshape(H, W * 3)], axis=1).tobytes()

    def chunk(kind: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main(samples: int = 30, scene: str = "final-scene", image_width: int = 200, aspect_ratio: float = 1.0,
         max_depth: int = 50, scene_seed: int = 1, seed: int = 0x5EED0001, device: int = 0,
         output: Optional[str] = None, devices: Optional[Sequence[int]] = None, split: str = "tile"):
    """(-main :samples n) on the GPU. Returns (argb (H, W) uint32 with row 0 = bottom, stats).
    devices: several GPUs of this node behind one call (rt_multi_*: what the reference's one thread per row and its
    latch become, core.clj:109-128), the frame split by tiles ("tile": bit-identical to one GPU) or by samples
    ("sample"); None or one id renders on `device` alone."""
    image_height = int(image_width / aspect_ratio)  # core.clj:54
    cfg, flat = setup(scene, scene_seed)
    cam = build_camera(cfg, aspect_ratio)
    params = abi.make_params(image_width, image_height, samples, max_depth, cfg["background"], seed)
    if split not in ("tile", "sample"):
        raise ValueError(f"split must be 'tile' or 'sample', not {split!r}")
    if devices is not None and len(devices) > 1:
        mctx = lib.MultiContext(list(devices))
        msc = mctx.scene(flat)
        try:
            _, argb, stats = msc.render(cam, params, abi.RT_SPLIT_TILES if split == "tile" else abi.RT_SPLIT_SAMPLES)
        finally:
            msc.close()
            mctx.close()
    else:
        ctx = lib.Context(devices[0] if devices else device)
        sc = ctx.scene(flat)
        try:
            _, argb, stats = sc.render(cam, params, want_rgb=False, want_argb=True)
        finally:
            sc.close()
            ctx.close()
    if output:
        (write_png if output.lower().endswith(".png") else write_ppm)(output, argb)
    return argb, stats


if __name__ == "__main__":  # python -m raytracing_amd.core --samples 10 [--scene cornell-box --width 400 --gpus 0,1 --output out.ppm]
    import argparse

    ap = argparse.ArgumentParser(description="(-main :samples n) of core.clj:50-51 on the GPU")
    ap.add_argument("--samples", type=int, default=30)
    ap.add_argument("--scene", default="final-scene", choices=sorted(scenes.SCENES))
    ap.add_argument("--width", type=int, default=200)
    ap.add_argument("--aspect", type=float, default=1.0)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--gpus", default="0", help="device ids, comma-separated; more than one renders through rt_multi_*")
    ap.add_argument("--split", default="tile", choices=["tile", "sample"])
    ap.add_argument("--output", default=None, help="write the frame to this file: a PNG if the name ends in .png, else a binary PPM")
    a = ap.parse_args()
    ids = [int(v) for v in a.gpus.split(",")]
    _, st = main(a.samples, a.scene, a.width, a.aspect, a.depth, devices=ids, split=a.split, output=a.output)
    print(f"{a.scene} {a.width}x{int(a.width / a.aspect)} @{a.samples}: {st['samples']} samples, {st['rays']} rays, kernel {st['kernel_ms']:.2f} ms")
