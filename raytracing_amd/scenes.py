"""The reference's eight world constructors, mirrored on the host.

Each returns the world as a Python list of hittables, as the Clojure constructors return a
vector. Construction-time randomness (`rand`, `Math/random`, `shuffle`) comes from the `rng`
argument (a `random.Random`), so a world is reproducible from a scene seed; the reference's own
stream (one process-global java.util.Random) cannot be reproduced.

Reference: src/raytracing/scenes.clj; per-scene camera overrides: core.clj:60-88.
"""
from __future__ import annotations

import math
import random
from typing import Callable, Dict, List

from .hittables import (box, bvhnode, constant_medium, hittable_list, moving_sphere, rotate_y,
                        sphere, translate, xy_rect, xz_rect, yz_rect)
from .materials import Metal, dielectric, diffuse_light, lambertian, metal
from .texture import checker_texture, image_texture, noise_texture, resource


def _rng(rng):
    return rng if rng is not None else random.Random(1)


def v_dist(v1, v2) -> float:  # scenes.clj:20-21 (sqrt of the length, as written)
    d = [a - b for a, b in zip(v1, v2)]
    return math.sqrt(math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))


def z_fix(objects: List) -> List:  # scenes.clj:23-26 (stable sort)
    return sorted(objects, key=lambda o: v_dist((13.0, 2.0, 3.0), o.center))


def two_spheres(rng=None) -> List:  # scenes.clj:28-32
    checker = checker_texture([0.2, 0.3, 0.1], [0.9, 0.9, 0.9])
    return z_fix([sphere([0, -10, 0], 10, lambertian(checker)),
                  sphere([0, 10, 0], 10, lambertian(checker))])


def two_perlin_spheres(rng=None) -> List:  # scenes.clj:34-37
    pertext = noise_texture(_rng(rng), 4.0)
    return z_fix([sphere([0, -1000, 0], 1000, lambertian(pertext)),
                  sphere([0, 2, 0], 2, lambertian(pertext))])


def earth(rng=None) -> List:  # scenes.clj:39-42
    earth_texture = image_texture(resource("earthmap.jpg"))
    earth_surface = lambertian(earth_texture)
    return [sphere([0, 0, 0], 2, earth_surface)]


def simple_light(rng=None) -> List:  # scenes.clj:44-49
    pertext = noise_texture(_rng(rng), 4)
    difflight = diffuse_light([4, 4, 4])
    return [sphere([0, -1000, 0], 1000, lambertian(pertext)),
            sphere([0, 2, 0], 2, lambertian(pertext)),
            xy_rect(3, 5, 1, 3, -2, difflight)]


def random_balls(rng=None) -> List:  # scenes.clj:51-61
    rng = _rng(rng)
    out = []
    for a in range(-11, 11):
        for b in range(-11, 11):
            center = [a + 0.9 * rng.random(), 0.2, b + 0.9 * rng.random()]
            choose_mat = rng.random()
            if choose_mat < 0.8:
                v1 = [rng.random() for _ in range(3)]
                v2 = [rng.random() for _ in range(3)]
                out.append(sphere(center, 0.2, lambertian([x * y for x, y in zip(v1, v2)])))
            elif choose_mat < 0.95:
                alb = [0.5 + (1 - 0.5) * rng.random() for _ in range(3)]
                out.append(sphere(center, 0.2, Metal(alb, 0 + (0.5 - 0) * rng.random())))
            else:
                out.append(sphere(center, 0.2, dielectric(1.5)))
    return out


def random_scene(rng=None) -> List:  # scenes.clj:63-69
    return z_fix(random_balls(rng) +
                 [sphere([4, 1, 0], 1.0, Metal([0.7, 0.6, 0.5], 0.0)),
                  sphere([0, 1, 0], 1.0, Metal([0.7, 0.6, 0.5], 0.0)),
                  sphere([-4, 1, 0], 1.0, Metal([0.7, 0.6, 0.5], 0.0)),
                  sphere([0, -1000, 0], 1000,
                         lambertian(checker_texture([0.2, 0.3, 0.1], [0.9, 0.9, 0.9]))),
                  sphere([0, -1000, 0], 1000, lambertian([0.5, 0.5, 0.5]))])


def cornell_box(rng=None) -> List:  # scenes.clj:71-87
    red = lambertian([0.65, 0.05, 0.05])
    white = lambertian([0.73, 0.73, 0.73])
    green = lambertian([0.12, 0.45, 0.15])
    light = diffuse_light([15, 15, 15])
    return [yz_rect(0, 555, 0, 555, 555, green),
            yz_rect(0, 555, 0, 555, 0, red),
            xz_rect(213, 343, 227, 332, 554, light),
            xz_rect(0, 555, 0, 555, 0, white),
            xz_rect(0, 555, 0, 555, 555, white),
            xy_rect(0, 555, 0, 555, 555, white),
            translate(rotate_y(box([0, 0, 0], [165, 330, 165], white), 15), [265, 0, 295]),
            translate(rotate_y(box([0, 0, 0], [165, 165, 165], white), -18), [130, 0, 65])]


def cornell_smoke(rng=None) -> List:  # scenes.clj:89-107
    red = lambertian([0.65, 0.05, 0.05])
    white = lambertian([0.73, 0.73, 0.73])
    green = lambertian([0.12, 0.45, 0.15])
    light = diffuse_light([7, 7, 7])
    box1 = translate(rotate_y(box([0, 0, 0], [165, 330, 165], white), 15), [265, 0, 295])
    box2 = translate(rotate_y(box([0, 0, 0], [165, 165, 165], white), -18), [130, 0, 65])
    return [yz_rect(0, 555, 0, 555, 555, green),
            yz_rect(0, 555, 0, 555, 0, red),
            xz_rect(113, 443, 127, 432, 554, light),
            xz_rect(0, 555, 0, 555, 555, white),
            xz_rect(0, 555, 0, 555, 0, white),
            xy_rect(0, 555, 0, 555, 555, white),
            constant_medium(box1, 0.01, [0.0, 0.0, 0.0]),
            constant_medium(box2, 0.01, [1.0, 1.0, 1.0])]


def final_scene(rng=None) -> List:  # scenes.clj:109-146
    rng = _rng(rng)
    ground = lambertian([0.48, 0.83, 0.53])
    boxes_per_side = 20
    boxes1 = []
    for i in range(boxes_per_side):
        for j in range(boxes_per_side):
            w = 100.0
            x0 = -1000.0 + i * w
            z0 = -1000.0 + j * w
            y0 = 0.0
            x1 = x0 + w
            y1 = 1.0 + (101.0 - 1.0) * rng.random()
            z1 = z0 + w
            boxes1.append(box([x0, y0, z0], [x1, y1, z1], ground))
    light = diffuse_light([7, 7, 7])
    center1 = [400, 400, 200]
    center2 = [center1[0] + 30, center1[1] + 0, center1[2] + 0]
    moving_sphere_material = lambertian([0.7, 0.3, 0.1])
    boundary = sphere([360, 150, 145], 70, dielectric(1.5))
    emat = lambertian(image_texture(resource("earthmap.jpg")))
    pertext = noise_texture(rng, 0.1)
    white = lambertian([0.73, 0.73, 0.73])
    boxes2 = [sphere([0 + (165 - 0) * rng.random() for _ in range(3)], 10, white) for _ in range(1000)]
    return [bvhnode(hittable_list(boxes1), 0, 1, rng),
            xz_rect(123, 423, 147, 412, 554, light),
            moving_sphere(center1, center2, 0, 1, 50, moving_sphere_material),
            sphere([260, 150, 45], 50, dielectric(1.5)),
            sphere([0, 150, 145], 50, metal([0.8, 0.8, 0.9], 1.0)),
            boundary,
            constant_medium(boundary, 0.2, [0.0, 0.4, 0.9]),
            constant_medium(sphere([0, 0, 0], 5000, dielectric(1.5)), 0.0001, [1, 1, 1]),
            sphere([400, 200, 400], 100, emat),
            sphere([220, 280, 300], 80, lambertian(pertext)),
            translate(rotate_y(bvhnode(hittable_list(boxes2), 0.0, 1.0, rng), 15), [-100, 270, 395])]


# core.clj:60-88: the scene table of -main with its :or defaults filled in.
_DEFAULTS = dict(lookfrom=(13.0, 2.0, 3.0), lookat=(0.0, 0.0, 0.0), aperture=0.0, vfov=20.0,
                 background=(0.70, 0.80, 1.0))

SCENES: Dict[str, dict] = {
    "random-scene": dict(world=random_scene, aperture=0.1),
    "two-spheres": dict(world=two_spheres),
    "two-perlin-spheres": dict(world=two_perlin_spheres),
    "earth": dict(world=earth),
    "simple-light": dict(world=simple_light, background=(0.0, 0.0, 0.0), lookfrom=(26.0, 3.0, 6.0),
                         lookat=(0.0, 2.0, 0.0)),
    "cornell-box": dict(world=cornell_box, background=(0.0, 0.0, 0.0), lookfrom=(278.0, 278.0, -800.0),
                        lookat=(278.0, 278.0, 0.0), vfov=40.0),
    "cornell-smoke": dict(world=cornell_smoke, lookfrom=(278.0, 278.0, -800.0),
                          lookat=(278.0, 278.0, 0.0), vfov=40.0),
    "final-scene": dict(world=final_scene, background=(0.0, 0.0, 0.0), lookfrom=(478.0, 278.0, -600.0),
                        lookat=(278.0, 278.0, 0.0), vfov=40.0),
}


def scene_config(name: str) -> dict:
    """The destructured map of core.clj:60-65 for scene keyword `name`."""
    if name not in SCENES:
        raise KeyError(f"unknown scene {name!r}; choose from {sorted(SCENES)}")
    cfg = dict(_DEFAULTS)
    cfg.update(SCENES[name])
    return cfg
