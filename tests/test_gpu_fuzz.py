"""Random worlds out of everything the path knows — spheres, moving spheres, rects, boxes, lists, rotate-y
and translate wrappers nested up to three deep, constant media with sphere, box and list boundaries,
all five materials, checker / noise / image textures — against the oracle: closest-hit records (with scripted
medium uniforms) and per-sample radiance with the shared random stream, on every kernel instantiation that
covers the world. The fixed worlds of the other tests were written with the device's fast paths in mind;
these are not."""
import os

import numpy as np
import pytest

from raytracing_amd import abi, flatten
from raytracing_amd.core import build_camera
from raytracing_amd.hittables import (box, constant_medium, hittable_list, moving_sphere, rotate_y, sphere, translate,
                                      xy_rect, xz_rect, yz_rect)
from raytracing_amd.materials import Metal, dielectric, diffuse_light, lambertian, metal
from raytracing_amd.texture import checker_texture, image_texture, noise_texture, resource, solid_color
from test_gpu_parity import _check_hits

pytestmark = pytest.mark.gpu


class _PyRng:  # the two calls texture.perlin makes
    def __init__(self, seed):
        self.r = np.random.default_rng(seed)

    def random(self):
        return float(self.r.random())

    def shuffle(self, x):
        self.r.shuffle(x)


PROFILES = {  # what a world may contain -> the instantiation it should land on
    "any": None,
    "rect": dict(shapes=(2, 3, 4, 5), mats=(0, 8), wrap=3, media=(1,), want=("V_RECT", "V_SMOKE", "V_ALL_DEEP")),
    "list": dict(shapes=(0, 1, 2, 3, 4), mats=(0, 5, 6, 7, 8, 9, 10), wrap=0, media=(), want=("V_LIST", "V_SPH", "V_BALLS")),
    "noise": dict(shapes=(0, 2, 3, 4), mats=(0, 8, 9, 11), wrap=0, media=(), want=("V_NOISE", "V_LIST", "V_RECT")),
    # big lists below wrappers: trees under instance chains (rects there are met in the chain's space, spheres in world space)
    "instanced": dict(shapes=(0, 1, 2, 3, 4, 5), mats=None, wrap=3, media=(0, 1), want=None, lists=(18, 70)),
}


def _material(rng, textures, allowed=None):
    k = int(rng.integers(0, 10)) if allowed is None else int(allowed[int(rng.integers(len(allowed)))])
    if k == 10:
        return lambertian(textures[1])      # the image
    if k == 11:
        return lambertian(textures[0])      # noise
    col = [float(c) for c in rng.uniform(0.1, 0.9, 3)]
    if k <= 3:
        return lambertian(col)
    if k == 4:
        return lambertian(textures[int(rng.integers(len(textures)))])
    if k == 5:
        return metal(col, float(rng.uniform(0, 0.6)))
    if k == 6:
        return Metal(col, float(rng.uniform(0, 1.5)))       # the record constructor does not clamp the fuzz
    if k == 7:
        return dielectric(float(rng.uniform(1.1, 2.0)))
    if k == 8:
        return diffuse_light([float(c) for c in rng.uniform(1, 6, 3)])
    return lambertian(checker_texture(col, [0.9, 0.9, 0.9]))


def _shape(rng, mat, extent, allowed=None):
    k = int(rng.integers(0, 7)) if allowed is None else int(allowed[int(rng.integers(len(allowed)))])
    c = rng.uniform(-extent, extent, 3)
    if k == 0:
        return sphere(list(c), float(rng.uniform(0.3, 2.5)), mat)
    if k == 1:
        return moving_sphere(list(c), list(c + rng.uniform(-1, 1, 3)), 0.0, 1.0, float(rng.uniform(0.3, 1.5)), mat)
    if k in (2, 3, 4):
        a0, b0 = c[0], c[1]
        a1, b1 = a0 + rng.uniform(0.5, 6), b0 + rng.uniform(0.5, 6)
        # `^long k` truncates c[2]. Not into the plane through the origin, though: a checker there takes its colour from
        # the sign of a rounding error (tests/test_gpu_textures.py), which no two arithmetics agree on sample by sample.
        kk = float(c[2]) if abs(c[2]) >= 1.0 else float(c[2]) + (1.0 if c[2] >= 0 else -1.0)
        return (xy_rect, xz_rect, yz_rect)[k - 2](a0, a1, b0, b1, kk, mat)
    if k == 5:
        # (no face in a coordinate plane through the origin either: box.clj hands its six k to `^long`)
        lo, hi = c, c + rng.uniform(0.5, 4, 3)
        off = lambda v: np.where(np.abs(v) >= 1.0, v, v + np.where(v >= 0, 1.0, -1.0))
        return box(list(off(lo)), list(np.maximum(off(hi), off(lo) + 0.5)), mat)
    return hittable_list([_shape(rng, mat, extent) for _ in range(int(rng.integers(2, 4)))])


def _wrapped(rng, obj, depth):
    for _ in range(depth):
        if rng.random() < 0.5:
            obj = translate(obj, [float(v) for v in rng.uniform(-3, 3, 3)])
        else:
            obj = rotate_y(obj, float(rng.uniform(-80, 80)))
    return obj


def _world(rng, n_objects, extent, profile=None):
    textures = [noise_texture(_PyRng(int(rng.integers(1 << 30))), float(rng.uniform(1, 6))),
                image_texture(resource("stones.jpg")),
                checker_texture(noise_texture(_PyRng(int(rng.integers(1 << 30))), 3.0), solid_color(0.2, 0.3, 0.1))]
    world = []
    for _ in range(n_objects):
        if profile is None:
            obj = _wrapped(rng, _shape(rng, _material(rng, textures), extent), int(rng.integers(0, 4)))
        elif "lists" in profile:
            members = [_shape(rng, _material(rng, textures), extent, profile["shapes"])
                       for _ in range(int(rng.integers(*profile["lists"])))]
            obj = _wrapped(rng, hittable_list(members), int(rng.integers(1, profile["wrap"] + 1)))
        else:
            obj = _wrapped(rng, _shape(rng, _material(rng, textures, profile["mats"]), extent, profile["shapes"]),
                           int(rng.integers(0, profile["wrap"] + 1)))
        if rng.random() < 0.12 and (profile is None or profile["media"]):
            kind = int(rng.integers(0, 3)) if profile is None else int(profile["media"][int(rng.integers(len(profile["media"])))])
            c = rng.uniform(-extent, extent, 3)
            if kind == 0:
                b = sphere(list(c), float(rng.uniform(1, 4)), lambertian([0.5, 0.5, 0.5]))
            elif kind == 1:
                if profile is not None and "lists" not in profile:     # whole coordinates: `^long k` then leaves the six rects a closed box
                    c = np.round(c)
                    b = box(list(c), list(c + rng.integers(1, 5, 3)), lambertian([0.5, 0.5, 0.5]))
                else:
                    b = box(list(c), list(c + rng.uniform(1, 5, 3)), lambertian([0.5, 0.5, 0.5]))
            else:
                b = hittable_list([sphere(list(c), 2.0, lambertian([0.5, 0.5, 0.5])),
                                   box(list(c), list(c + 2.5), lambertian([0.5, 0.5, 0.5]))])
            obj = constant_medium(_wrapped(rng, b, int(rng.integers(0, 3))), float(rng.uniform(0.05, 0.8)),
                                  [float(v) for v in rng.uniform(0.1, 0.9, 3)])
        world.append(obj)
    world.append(xz_rect(-3 * extent, 3 * extent, -3 * extent, 3 * extent, -extent - 1, lambertian([0.5, 0.6, 0.5])))
    return world


CASES = [("any", s, n) for s, n in zip(range(16), [4, 6, 9, 12, 14, 20, 30, 45, 5, 8, 60, 16, 25, 7, 90, 11])]
CASES += [("rect", 20 + s, n) for s, n in zip(range(6), [3, 5, 8, 4, 6, 10])]
CASES += [("list", 30 + s, n) for s, n in zip(range(8), [3, 6, 10, 14, 30, 80, 200, 12])]
CASES += [("noise", 40 + s, n) for s, n in zip(range(5), [2, 4, 8, 12, 15])]
CASES += [("instanced", 50 + s, n) for s, n in zip(range(6), [2, 2, 3, 4, 6, 9])]
if os.environ.get("RT_FUZZ_SWEEP"):    # a wider sweep on demand, RT_FUZZ_SWEEP=<first seed>:<count>, other worlds than the suite's
    _first, _count = (int(v) for v in os.environ["RT_FUZZ_SWEEP"].split(":"))
    _sizes = [4, 6, 9, 12, 14, 20, 30, 45, 5, 8, 60, 16, 25, 7, 90, 11, 3, 200, 2, 35]
    _profiles = ["any", "any", "rect", "list", "noise", "instanced"]
    CASES = [(_profiles[s % 6], s, min(_sizes[s % 20], 9 if _profiles[s % 6] == "instanced" else 10 if _profiles[s % 6] == "rect" else 15 if _profiles[s % 6] == "noise" else 200))
             for s in range(_first, _first + _count)]


@pytest.mark.parametrize("profile,seed,n_objects", CASES)
def test_random_world(gpu_ctx, oracle, profile, seed, n_objects):
    _random_world(gpu_ctx, oracle, profile, seed, n_objects, 3e-4)


@pytest.mark.parametrize("profile,seed,n_objects", [("any", 5797, 200), ("any", 7194, 90)])
def test_two_worlds_of_the_wide_sweep_that_overran_the_budget(gpu_ctx, oracle, profile, seed, n_objects):
    """The two worlds of a 1 500-world sweep (RT_FUZZ_SWEEP) whose knife-edge records came to one more than the suite's
    budget of 2 in 8 000: 3 records each, rays that graze a rect's edge below a rotate-y wrapper, where the FP32
    chain-space ray and the FP64 one fall on different sides. Kept here at a budget of 4 so that they stay watched."""
    _random_world(gpu_ctx, oracle, profile, seed, n_objects, 5e-4)


@pytest.mark.parametrize("profile,seed,n_objects", [("instanced", 40397, 9), ("instanced", 41705, 9), ("instanced", 42209, 8)])
def test_three_worlds_of_a_wider_sweep_that_overran_the_budget(gpu_ctx, oracle, profile, seed, n_objects):
    """The three worlds of a 4 000-world sweep (seeds 40000 on) that overran: one with 3 knife-edge records of 8 000,
    two with 12 of 3 000 samples whose radiance diverges (budget 9). All three are trees of rects below wrappers, the
    one kind of world where a device ray (FP32 in the wrapper's space) and the reference's (FP64) part ways often enough
    to count: over 200 such worlds the diverged samples per world were 0 in 77, 1 in 50, 2 in 43, 3 in 17, 4 to 6 in 11
    and 9 and 12 in one each, against 0 (seldom 1 or 2) in every world of the other kinds. A sample diverges when
    somewhere along its path one of the two rays passes a rect's edge and the other does not; both are paths the
    reference could have taken. Kept here at 4 records and 15 samples so that they stay watched."""
    _random_world(gpu_ctx, oracle, profile, seed, n_objects, 5e-4, diverged_budget=15)


def _random_world(gpu_ctx, oracle, profile, seed, n_objects, knife, diverged_budget=None):
    rng = np.random.default_rng(7000 + seed)
    extent = 6.0 + 0.15 * min(n_objects, 100)
    flat = flatten.flatten_world(_world(rng, n_objects, extent, PROFILES[profile]))
    sc = gpu_ctx.scene(flat)
    rv, cover = sc.variants()
    if PROFILES[profile] and PROFILES[profile]["want"]:
        assert abi.VARIANT_NAMES[rv] in PROFILES[profile]["want"], abi.VARIANT_NAMES[rv]
    nm = flat.desc.n_media
    n = 8000
    o = rng.uniform(-1.6 * extent, 1.6 * extent, (n, 3)).astype(np.float32)
    d = (rng.uniform(-extent, extent, (n, 3)) - o).astype(np.float32)
    d[: n // 4] = rng.normal(size=(n // 4, 3)).astype(np.float32)
    tm = rng.random(n).astype(np.float32)
    xi = rng.random((n, nm)).astype(np.float32) if nm else None
    dev = sc.hit_batch(o, d, tm, medium_xi=xi)
    orc = oracle.hit_batch(flat, o, d, tm, medium_xi=xi)
    assert int(orc[:, 11].sum()) > n // 10
    _check_hits(dev, orc, o, d, knife, label=f"random world {seed} ({n_objects} objects, {flat.desc.n_prims} primitives, "
                                             f"{nm} media, renders on {abi.VARIANT_NAMES[rv]})",
                requery=lambda oo: oracle.hit_batch(flat, oo, d, tm, medium_xi=xi))
    for v in cover:
        assert sc.hit_batch(o, d, tm, medium_xi=xi, variant=v).tobytes() == dev.tobytes(), abi.VARIANT_NAMES[v]
    # the same rays over a window of t that cuts through the world (the t-min / t-max of base.clj:92-101)
    lo, hi = float(rng.uniform(1e-3, 0.2 * extent)) / float(np.median(np.linalg.norm(d, axis=1))), float(rng.uniform(0.6, 1.6))
    dev_w = sc.hit_batch(o, d, tm, t_min=lo, t_max=hi, medium_xi=xi)
    orc_w = oracle.hit_batch(flat, o, d, tm, t_min=lo, t_max=hi, medium_xi=xi)
    _check_hits(dev_w, orc_w, o, d, knife, label=f"random world {seed}, t in [{lo:.3g}, {hi:.3g}]",
                requery=lambda oo: oracle.hit_batch(flat, oo, d, tm, t_min=lo, t_max=hi, medium_xi=xi))
    # per-sample radiance through the instantiation the render uses
    cfg = {"lookfrom": [0.0, 0.6 * extent, -2.2 * extent], "lookat": [0.0, 0.0, 0.0], "vfov": 50.0, "aperture": 0.1,
           "background": [0.5, 0.6, 0.8]}
    cam = build_camera(cfg, 1.0)
    W = H = 48
    p = abi.make_params(W, H, 4, 50, cfg["background"], seed=11 + seed)
    m = 3000
    px = rng.integers(0, W * H, m).astype(np.int32)
    smp = rng.integers(0, 2000, m).astype(np.int32)
    a = sc.sample_batch(cam, p, px, smp)
    b = oracle.sample_batch(flat, cam, p, px, smp)
    rel = np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-2)
    diverged = int((rel > 2e-3).sum())
    print(f"[parity] random world {seed} per-sample radiance: {diverged} of {m} diverged, median {np.median(rel):.2e}")
    assert diverged <= (diverged_budget or max(3, int(3e-3 * m))) and np.median(rel) <= 2e-5
    # and a small frame equals the ordered sum of its samples
    q = abi.make_params(24, 24, 3, 50, cfg["background"], seed=11 + seed)
    rgb, _, st = sc.render(cam, q)
    px2 = np.repeat(np.arange(24 * 24, dtype=np.int32), 3)
    smp2 = np.tile(np.arange(3, dtype=np.int32), 24 * 24)
    s = sc.sample_batch(cam, q, px2, smp2).reshape(24 * 24, 3, 3).astype(np.float32)
    want = np.zeros((24 * 24, 3), np.float32)
    for k in range(3):
        want = want + s[:, k, :]
    assert np.array_equal(rgb.astype(np.float32).reshape(-1, 3).view(np.uint32), want.view(np.uint32))
    # the same frame from the kernel that re-packs paths (where the world can use it; ignored elsewhere), from two
    # tile shards added up, and from the world written to a scene file and read back
    bits = rgb.view(np.uint32)
    pool, _, st2 = sc.render(cam, abi.make_params(24, 24, 3, 50, cfg["background"], seed=11 + seed, flags=abi.RT_FLAG_POOL_PATHS))
    assert np.array_equal(pool.view(np.uint32), bits) and st2["rays"] == st["rays"]
    halves = [sc.render(cam, abi.make_params(24, 24, 3, 50, cfg["background"], seed=11 + seed, shard_index=k, shard_count=2))[0]
              for k in (0, 1)]
    assert np.array_equal((halves[0] + halves[1]).view(np.uint32), bits)
    from raytracing_amd import scenefile
    again = gpu_ctx.scene_from_bytes(scenefile.dumps(flat))
    assert np.array_equal(again.render(cam, q)[0].view(np.uint32), bits)
    assert again.hit_batch(o, d, tm, medium_xi=xi).tobytes() == dev.tobytes()
    again.close()
    sc.close()
