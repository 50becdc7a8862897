"""Host-side mirror of the reference's textures and Perlin tables (no evaluation here).

Reference: src/raytracing/texture.clj, noise.clj.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Any, List, Optional, Tuple

import numpy as np


@dataclass(eq=False)
class SolidColor:  # texture.clj:15
    color_value: Tuple[float, float, float]


def solid_color(*c) -> SolidColor:  # texture.clj:19-21
    if len(c) == 1:
        c = c[0]
    return SolidColor((float(c[0]), float(c[1]), float(c[2])))


def is_texture(a) -> bool:  # (satisfies? Texture a)
    return isinstance(a, (SolidColor, CheckerTexture, NoiseTexture, ImageTexture))


@dataclass(eq=False)
class CheckerTexture:  # texture.clj:23
    odd: Any
    even: Any


def checker_texture(c1, c2) -> CheckerTexture:  # texture.clj:32-36
    if is_texture(c1):
        return CheckerTexture(c1, c2)
    return CheckerTexture(solid_color(c1), solid_color(c2))


@dataclass(eq=False)
class Perlin:  # noise.clj:35-37
    point_count: int
    ranvec: np.ndarray   # (256, 3) float64
    perm_x: np.ndarray   # (256,) int32
    perm_y: np.ndarray
    perm_z: np.ndarray


def perlin(rng) -> Perlin:  # noise.clj:72-83; rng.random() stands in for Math/random, rng.shuffle for shuffle
    point_count = 256
    ranvec = np.empty((point_count, 3), dtype=np.float64)
    for i in range(point_count):
        v = [-1.0 + 2.0 * rng.random() for _ in range(3)]  # random-vec3 -1 1
        ln = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        inv = 1.0 / ln  # unit-vector = vd v (length v)
        ranvec[i] = (v[0] * inv, v[1] * inv, v[2] * inv)

    def perm():
        p = list(range(256))
        rng.shuffle(p)
        return np.asarray(p, dtype=np.int32)
    return Perlin(point_count, ranvec, perm(), perm(), perm())


@dataclass(eq=False)
class NoiseTexture:  # texture.clj:38-39 (scale is a ^long field)
    perlin_noise: Perlin
    scale: int


def noise_texture(rng, sc=1.0) -> NoiseTexture:  # texture.clj:46-48
    return NoiseTexture(perlin(rng), int(sc))


@dataclass(eq=False)
class ImageTexture:  # texture.clj:50-53
    data: Optional[np.ndarray]  # (h, w, 3) uint8, top row first, R G B; None = the no-data texture
    width: int
    height: int


ASSET_DIR = os.path.join(os.path.dirname(__file__), "assets")


def load_rgb8(path: str) -> np.ndarray:
    """Decode an image file to (h, w, 3) uint8, as ImageIO/read does for the reference (texture.clj:72).
    JPEG, PNG, GIF and BMP files (told apart by their first bytes, whatever the name: resources/map2.png is a JPEG)
    are decoded by the library's own decoders, rt_image_decode; `.npz` files (key 'rgb8') hold decoded arrays."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return np.ascontiguousarray(z["rgb8"], dtype=np.uint8)
    with open(path, "rb") as f:
        data = f.read()
    from . import lib
    return lib.decode_image(data)


def image_texture(source=None) -> ImageTexture:  # texture.clj:68-81
    if source is None:
        return ImageTexture(None, 0, 0)
    arr = load_rgb8(source) if isinstance(source, str) else np.ascontiguousarray(source, dtype=np.uint8)
    if arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError("image texture needs an (h, w, 3) RGB8 array")
    return ImageTexture(arr, int(arr.shape[1]), int(arr.shape[0]))


def resource(name: str) -> str:
    """Stand-in for (io/resource name): the file of that name shipped with this package."""
    return os.path.join(ASSET_DIR, name)
