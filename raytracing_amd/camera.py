"""The camera map of the reference (src/raytracing/camera.clj:7-33), built once per render on
the host in FP64 and handed to the device by value."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Tuple

Vec = Tuple[float, float, float]


def _sub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def _mul(a, k): return (a[0] * k, a[1] * k, a[2] * k)
def _cross(u, v): return (u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0])


def _unit(v):  # base.clj:39-40: multiply by the reciprocal of the length
    return _mul(v, 1.0 / math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))


@dataclass
class Camera:
    origin: Vec
    lower_left_corner: Vec
    horizontal: Vec
    vertical: Vec
    u: Vec
    v: Vec
    w: Vec
    lens_radius: float
    time0: float
    time1: float


def camera(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time0, time1) -> Camera:
    lookfrom = tuple(float(x) for x in lookfrom)
    lookat = tuple(float(x) for x in lookat)
    vup = tuple(float(x) for x in vup)
    theta = vfov * 0.017453292519943295  # Math/toRadians
    h = math.tan(theta * 0.5)
    viewport_height = h * 2.0
    viewport_width = aspect_ratio * viewport_height
    w = _unit(_sub(lookfrom, lookat))
    u = _unit(_cross(vup, w))
    v = _cross(w, u)
    origin = lookfrom
    horizontal = _mul(u, focus_dist * viewport_width)
    vertical = _mul(v, focus_dist * viewport_height)
    llc = _sub(_sub(_sub(origin, _mul(horizontal, 1.0 / 2)), _mul(vertical, 1.0 / 2)), _mul(w, focus_dist))
    return Camera(origin, llc, horizontal, vertical, u, v, w, aperture / 2, float(time0), float(time1))
