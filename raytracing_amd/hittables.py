"""Host-side mirror of the reference's Hittable constructors.

These classes hold exactly the fields the reference's records hold, so that
`flatten.flatten_world` can walk them the way a Clojure flattener would walk
`(:objects …)`, `(:left …)`, `(:ptr …)`, `(:boundary …)` and `(:sides …)`.
Nothing here intersects rays: `hit` lives on the GPU.

Reference: src/raytracing/base.clj, rectangles.clj, box.clj, rotate.clj,
translate.clj, constant_medium.clj.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence, Tuple

Vec = Tuple[float, float, float]


def _v(x) -> Vec:
    return (float(x[0]), float(x[1]), float(x[2]))


def _long(x) -> float:
    """Coercion of a record field declared ^long (clojure.lang.RT/longCast truncates toward 0)."""
    return float(int(x))


@dataclass(eq=False)
class Aabb:  # base.clj:59
    minimum: Vec
    maximum: Vec


def surrounding_box(b0: Aabb, b1: Aabb) -> Aabb:  # base.clj:81-88
    return Aabb(tuple(min(a, b) for a, b in zip(b0.minimum, b1.minimum)),
                tuple(max(a, b) for a, b in zip(b0.maximum, b1.maximum)))


@dataclass(eq=False)
class Sphere:  # base.clj:229
    center: Vec
    radius: float
    mat_ptr: Any

    def bounding_box(self, time0, time1) -> Aabb:  # base.clj:253-257
        c, r = self.center, self.radius
        return Aabb((c[0] - r, c[1] - r, c[2] - r), (c[0] + r, c[1] + r, c[2] + r))


def sphere(center, radius, mat_ptr) -> Sphere:  # base.clj:260-261
    return Sphere(_v(center), float(radius), mat_ptr)


@dataclass(eq=False)
class MovingSphere:  # base.clj:263-267
    center0: Vec
    center1: Vec
    time0: float
    time1: float
    radius: float
    mat_ptr: Any

    def center(self, time_) -> Vec:  # base.clj:298-302
        k = (time_ - self.time0) / (self.time1 - self.time0)
        return tuple(a + (b - a) * k for a, b in zip(self.center0, self.center1))

    def bounding_box(self, time0, time1) -> Aabb:  # base.clj:293-297
        r = self.radius

        def bx(c):
            return Aabb((c[0] - r, c[1] - r, c[2] - r), (c[0] + r, c[1] + r, c[2] + r))
        return surrounding_box(bx(self.center(time0)), bx(self.center(time1)))


def moving_sphere(center0, center1, time0, time1, radius, mat_ptr) -> MovingSphere:  # base.clj:304-305
    return MovingSphere(_v(center0), _v(center1), float(time0), float(time1), float(radius), mat_ptr)


@dataclass(eq=False)
class XYRect:  # rectangles.clj:8-13 (k is a ^long field)
    mp: Any
    x0: float
    x1: float
    y0: float
    y1: float
    k: float

    def bounding_box(self, time0, time1) -> Aabb:  # rectangles.clj:38-40
        return Aabb((self.x0, self.y0, self.k - 0.0001), (self.x1, self.y1, self.k + 0.0001))


@dataclass(eq=False)
class XZRect:  # rectangles.clj:45-50
    mp: Any
    x0: float
    x1: float
    z0: float
    z1: float
    k: float

    def bounding_box(self, time0, time1) -> Aabb:  # rectangles.clj:75-77
        return Aabb((self.x0, self.k - 0.0001, self.z0), (self.x1, self.k + 0.0001, self.z1))


@dataclass(eq=False)
class YZRect:  # rectangles.clj:82-87
    mp: Any
    y0: float
    y1: float
    z0: float
    z1: float
    k: float

    def bounding_box(self, time0, time1) -> Aabb:  # rectangles.clj:112-114
        return Aabb((self.k - 0.0001, self.y0, self.z0), (self.k + 0.0001, self.y1, self.z1))


def xy_rect(x0, x1, y0, y1, k, mat) -> XYRect:  # rectangles.clj:42-43
    return XYRect(mat, float(x0), float(x1), float(y0), float(y1), _long(k))


def xz_rect(x0, x1, z0, z1, k, mat) -> XZRect:  # rectangles.clj:79-80
    return XZRect(mat, float(x0), float(x1), float(z0), float(z1), _long(k))


def yz_rect(y0, y1, z0, z1, k, mat) -> YZRect:  # rectangles.clj:116-117
    return YZRect(mat, float(y0), float(y1), float(z0), float(z1), _long(k))


@dataclass(eq=False)
class HittableList:  # base.clj:90
    objects: List[Any]


def hittable_list(objects) -> HittableList:  # base.clj:113-114
    return HittableList(list(objects))


@dataclass(eq=False)
class Box:  # box.clj:8
    box_min: Vec
    box_max: Vec
    sides: List[Any]

    def bounding_box(self, time0, time1) -> Aabb:  # box.clj:12-13
        return Aabb(self.box_min, self.box_max)


def box(p0, p1, ptr) -> Box:  # box.clj:15-24 (side order matters only for exact ties)
    x0, y0, z0 = p0
    x1, y1, z1 = p1
    sides = [xy_rect(x0, x1, y0, y1, z1, ptr),
             xy_rect(x0, x1, y0, y1, z0, ptr),
             xz_rect(x0, x1, z0, z1, y1, ptr),
             xz_rect(x0, x1, z0, z1, y0, ptr),
             yz_rect(y0, y1, z0, z1, x1, ptr),
             yz_rect(y0, y1, z0, z1, x0, ptr)]
    return Box(_v(p0), _v(p1), sides)


@dataclass(eq=False)
class BvhNode:  # base.clj:116
    left: Any
    right: Any
    box: Aabb

    def bounding_box(self, time0, time1) -> Aabb:  # base.clj:123-124
        return self.box


def bvhnode(list_or_objects, time0, time1, rng, start: Optional[int] = None, end: Optional[int] = None):
    """base.clj:137-164, including its index handling: every level re-sorts the WHOLE vector
    along its own random axis (0 or 1) and then recurses on index sub-ranges of that vector, so
    objects can be dropped or duplicated when the axes of two levels differ. `rng.random()`
    stands in for clojure.core/rand."""
    if isinstance(list_or_objects, HittableList):
        objects = list(list_or_objects.objects)
        start, end = 0, len(objects)
    else:
        objects = list(list_or_objects)
    axis = int(rng.random() * 2)  # (rand-int 2)

    def key(o):
        return o.bounding_box(0, 0).minimum[axis]
    span = end - start
    if span == 1:
        left = right = objects[start]
    elif span == 2:
        a, b = objects[start], objects[start + 1]
        left, right = (a, b) if key(a) < key(b) else (b, a)
    else:
        sorted_obj = sorted(objects, key=key)  # sort-by is stable, like sorted()
        mid = int(math.ceil(start + span / 2))
        left = bvhnode(sorted_obj, time0, time1, rng, start, mid)
        right = bvhnode(sorted_obj, time0, time1, rng, mid, end)
    return BvhNode(left, right, surrounding_box(left.bounding_box(time0, time1),
                                                right.bounding_box(time0, time1)))


@dataclass(eq=False)
class RotateY:  # rotate.clj:9 (hasbox/bbox are never read by `hit`; not mirrored)
    ptr: Any
    sin_theta: float
    cos_theta: float


def rotate_y(p, angle) -> RotateY:  # rotate.clj:51-54,76-80
    radians = math.radians(angle)
    return RotateY(p, math.sin(radians), math.cos(radians))


@dataclass(eq=False)
class Translate:  # translate.clj:8
    ptr: Any
    offset: Vec


def translate(p, displacement) -> Translate:  # translate.clj:23-24
    return Translate(p, _v(displacement))


@dataclass(eq=False)
class ConstantMedium:  # constant_medium.clj:9
    boundary: Any
    phase_function: Any
    neg_inv_density: float


def constant_medium(b, d, a) -> ConstantMedium:  # constant_medium.clj:48-53
    from .materials import isotropic
    return ConstantMedium(b, isotropic(a), -1.0 / float(d))
