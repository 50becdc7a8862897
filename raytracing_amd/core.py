"""Host-side mirror of the reference's entry point (src/raytracing/core.clj:50-129).

`main(samples=n)` keeps the meaning of `(-main :samples n)`: the constants of core.clj:52-59
(200x200, aspect 1.0, depth 50, :final-scene) are the defaults, and the sample loop of
core.clj:96-127 runs on the GPU through librtmi.so. The other keywords are what the reference
reaches only by editing its source.
"""
from __future__ import annotations

import random
from typing import Optional

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


def main(samples: int = 30, scene: str = "final-scene", image_width: int = 200, aspect_ratio: float = 1.0,
         max_depth: int = 50, scene_seed: int = 1, seed: int = 0x5EED0001, device: int = 0,
         output: Optional[str] = None):
    """(-main :samples n) on the GPU. Returns (argb (H, W) uint32 with row 0 = bottom, stats)."""
    image_height = int(image_width / aspect_ratio)  # core.clj:54
    cfg, flat = setup(scene, scene_seed)
    cam = build_camera(cfg, aspect_ratio)
    params = abi.make_params(image_width, image_height, samples, max_depth, cfg["background"], seed)
    ctx = lib.Context(device)
    sc = ctx.scene(flat)
    try:
        _, argb, stats = sc.render(cam, params, want_rgb=False, want_argb=True)
    finally:
        sc.close()
        ctx.close()
    if output:
        write_ppm(output, argb)
    return argb, stats
