"""Host-side mirror of the reference's materials (fields only; scatter runs on the GPU).

Reference: src/raytracing/materials.clj.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Tuple

from .texture import is_texture, solid_color


@dataclass(eq=False)
class Lambertian:  # materials.clj:14
    albedo: Any


def lambertian(a) -> Lambertian:  # materials.clj:26-29
    return Lambertian(a if is_texture(a) else solid_color(a))


@dataclass(eq=False)
class Metal:  # materials.clj:31 — scenes.clj:60,65-67 call this constructor directly (no clamp)
    albedo: Tuple[float, float, float]
    fuzz: float

    def __post_init__(self):
        self.albedo = (float(self.albedo[0]), float(self.albedo[1]), float(self.albedo[2]))
        self.fuzz = float(self.fuzz)


def metal(color, f) -> Metal:  # materials.clj:42-43
    return Metal(color, f if 1 > f else 1)


@dataclass(eq=False)
class Dielectric:  # materials.clj:45
    ir: float


def dielectric(ir) -> Dielectric:  # materials.clj:64-65
    return Dielectric(float(ir))


@dataclass(eq=False)
class DiffuseLight:  # materials.clj:67
    emit: Any


def diffuse_light(a) -> DiffuseLight:  # materials.clj:80-83
    return DiffuseLight(a if is_texture(a) else solid_color(a))


@dataclass(eq=False)
class Isotropic:  # materials.clj:85
    albedo: Any


def isotropic(c) -> Isotropic:  # materials.clj:93-96
    return Isotropic(c if is_texture(c) else solid_color(c))
