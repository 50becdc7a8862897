"""Boxes (box.clj:7-20) on the device are one slab test instead of six rect tests. These tests
aim at where the two formulations could part: rays that start inside a box or on a face, rays
along the axes, rays through edges and corners (exact ties between faces), boxes sharing a group
with loose rects, and six rects that only look like a box (two materials: not merged)."""
import os

import numpy as np
import pytest

from raytracing_amd import flatten
from raytracing_amd.hittables import (box, constant_medium, hittable_list, rotate_y, translate, xy_rect, xz_rect,
                                      yz_rect)
from raytracing_amd.materials import lambertian
from test_gpu_parity import _check_hits

pytestmark = pytest.mark.gpu
RED, GREEN, WHITE = lambertian([0.6, 0.1, 0.1]), lambertian([0.1, 0.6, 0.1]), lambertian([0.7, 0.7, 0.7])


def _world():
    two_tone = hittable_list([xy_rect(20, 24, 0, 4, 4, RED), xy_rect(20, 24, 0, 4, 0, RED),
                              xz_rect(20, 24, 0, 4, 4, RED), xz_rect(20, 24, 0, 4, 0, RED),
                              yz_rect(0, 4, 0, 4, 24, RED), yz_rect(0, 4, 0, 4, 20, GREEN)])
    return [box([0, 0, 0], [4, 4, 4], WHITE),                       # shares the identity group with loose rects
            box([6, 1, -2], [9, 3, 5], RED),                        # a second box in the same group
            xz_rect(-10, 30, -10, 30, -1, GREEN),                   # floor below everything
            xy_rect(0, 4, 0, 4, 4, GREEN),                          # coincides with the first box's z = 4 face
            two_tone,                                               # six rects, two materials
            translate(rotate_y(box([0, 0, 0], [3, 5, 3], WHITE), 25), [12, 0, 3]),
            rotate_y(translate(box([0, 0, 0], [2, 2, 2], RED), [1, 6, 1]), -40)]


def _rays(rng, n):
    o = rng.uniform(-6, 28, (n, 3))
    d = rng.normal(size=(n, 3))
    k = n // 4
    o[:k] = rng.uniform(0.2, 3.8, (k, 3))                           # inside the first box
    o[k:2 * k] = np.array([7.5, 2.0, 1.0]) + rng.uniform(-1, 1, (k, 3))  # inside the second
    axis = rng.integers(0, 3, n // 8)                               # exactly along an axis
    d[2 * k:2 * k + n // 8] = 0
    d[np.arange(2 * k, 2 * k + n // 8), axis] = rng.choice([-1.0, 1.0], n // 8)
    return o.astype(np.float32), d.astype(np.float32)


def test_boxes_against_the_rect_by_rect_oracle(gpu_ctx, oracle):
    flat = flatten.flatten_world(_world())
    sc = gpu_ctx.scene(flat)
    o, d = _rays(np.random.default_rng(41), 40000)
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    assert int(orc[:, 11].sum()) > 20000
    _check_hits(dev, orc, o, d, 1e-4)
    sc.close()


def test_edges_corners_and_faces_exactly(gpu_ctx, oracle):
    """Lattice targets approached along directions whose components are powers of two: the
    device's reciprocals, products and sums are then exact in FP32 as they are in FP64, so ties
    between faces are real ties on both sides and must fall the same way (the later face in scan
    order, and the later box)."""
    flat = flatten.flatten_world([box([0, 0, 0], [4, 4, 4], WHITE), box([4, 0, 0], [8, 4, 4], RED)])
    sc = gpu_ctx.scene(flat)
    targets = [(x, y, z) for x in (0, 2, 4, 6, 8) for y in (0, 2, 4) for z in (0, 2, 4)]
    dirs = [(sx * a, sy * b, sz * c) for (a, b, c) in ((4, 4, 4), (8, 4, 2), (2, 8, 4), (4, 2, 8), (16, 1, 1))
            for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    o, d = [], []
    for tg in targets:
        for dv in dirs:
            for back in (1.0, 0.25):      # from outside, and from near or inside the boxes
                o.append(np.subtract(tg, np.multiply(dv, back)))
                d.append(dv)
    o, d = np.float32(o), np.float32(d)
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    assert _check_hits(dev, orc, o, d, 0.0) <= 1
    same = (dev["hit"] == orc[:, 11].astype(int))
    both = same & (dev["hit"] == 1)
    assert same.all()
    assert np.array_equal(dev["t"][both], orc[both, 0].astype(np.float32))
    assert np.array_equal(dev["normal"][both], orc[both, 4:7].astype(np.float32))
    assert np.array_equal(dev["mat_id"][both], orc[both, 10].astype(np.int32))
    sc.close()


def _room(missing, order=("x1", "x0", "y0", "y1", "z1", "z0")):
    """Five rects that are faces of the box [0,8] x [0,6] x [0,10] (the Cornell room of scenes.clj:76-81 has
    z = 0 open), each wall with its own material, plus a light below the ceiling and a block inside."""
    faces = {"z0": xy_rect(0, 8, 0, 6, 0, WHITE), "z1": xy_rect(0, 8, 0, 6, 10, WHITE),
             "y0": xz_rect(0, 8, 0, 10, 0, WHITE), "y1": xz_rect(0, 8, 0, 10, 6, RED),
             "x0": yz_rect(0, 6, 0, 10, 0, RED), "x1": yz_rect(0, 6, 0, 10, 8, GREEN)}
    walls = [faces[k] for k in order if k != missing]
    return walls[:2] + [xz_rect(3, 5, 4, 6, 5.9, WHITE)] + walls[2:] + [translate(rotate_y(box([0, 0, 0], [2, 3, 2], WHITE), 20), [3, 0, 4])]


@pytest.mark.parametrize("missing", ["z0", "z1", "y0", "y1", "x0", "x1"])
def test_room_with_one_face_open_against_the_rect_by_rect_oracle(gpu_ctx, oracle, missing):
    """Five faces of one box are intersected as one slab test with the missing face marked: rays from
    inside, from outside through the opening, from outside onto the backs of the walls, along the axes."""
    flat = flatten.flatten_world(_room(missing))
    sc = gpu_ctx.scene(flat)
    rng = np.random.default_rng(3)
    n = 40000
    o = rng.uniform(-6, 16, (n, 3))
    o[: n // 2] = rng.uniform(0.1, 5.9, (n // 2, 3)) * np.array([8 / 6, 1.0, 10 / 6])   # inside the room
    d = rng.normal(size=(n, 3))
    k = n // 8
    d[n // 2: n // 2 + k] = 0
    d[np.arange(n // 2, n // 2 + k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    o, d = o.astype(np.float32), d.astype(np.float32)
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    assert int(orc[:, 11].sum()) > n // 3
    assert int((orc[: n // 2, 11] < 0.5).sum()) > 500       # rays that leave through the opening
    _check_hits(dev, orc, o, d, 1e-4)
    sc.close()


def test_room_lattice_rays_exactly(gpu_ctx, oracle):
    """Corners and edges of the open room approached along power-of-two directions: exact arithmetic on
    both sides, so ties between walls must fall the same way. Targets on the rim of the opening are
    left out (there the missing plane ties with a wall's). The walls are listed XY, XZ, YZ: the device scans
    rects type by type, so only in that order are ties ACROSS orientations the reference's (the later wall wins)."""
    room = _room("z0", order=("z1", "z0", "y0", "y1", "x0", "x1"))
    flat = flatten.flatten_world(room[:2] + room[3:6])     # the five walls alone
    sc = gpu_ctx.scene(flat)
    targets = [(x, y, z) for x in (0, 4, 8) for y in (0, 2, 6) for z in (2, 6, 10)]
    dirs = [(sx * a, sy * b, sz * c) for (a, b, c) in ((4, 4, 4), (8, 4, 2), (2, 8, 4), (4, 2, 8), (16, 1, 1))
            for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)]
    o, d = [], []
    for tg in targets:
        for dv in dirs:
            for back in (1.0, 0.125):
                o.append(np.subtract(tg, np.multiply(dv, back)))
                d.append(dv)
    o, d = np.float64(o), np.float64(d)
    # rays that cross the plane of the opening exactly on its rim are left out (exact in FP64: lattice values)
    tz = -o[:, 2] / d[:, 2]
    x, y = o[:, 0] + tz * d[:, 0], o[:, 1] + tz * d[:, 1]
    rim = ((x == 0) | (x == 8) | (y == 0) | (y == 6)) & (x >= 0) & (x <= 8) & (y >= 0) & (y <= 6)
    o, d = np.float32(o[~rim]), np.float32(d[~rim])
    assert len(o) > 1500
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    same = (dev["hit"] == orc[:, 11].astype(int))
    both = same & (dev["hit"] == 1)
    assert same.all()
    assert int(both.sum()) > 1000
    assert np.array_equal(dev["t"][both], orc[both, 0].astype(np.float32))
    assert np.array_equal(dev["normal"][both], orc[both, 4:7].astype(np.float32))
    assert np.array_equal(dev["mat_id"][both], orc[both, 10].astype(np.int32))
    sc.close()


def test_box_as_medium_boundary(gpu_ctx, oracle):
    """constant_medium.clj:11-20 asks its boundary for the entry over (-inf, inf) and for the exit
    after it: both come out of the same slab test."""
    smoke = constant_medium(translate(rotate_y(box([0, 0, 0], [4, 6, 4], WHITE), 15), [1, 0, 2]), 0.3, [0.2, 0.2, 0.2])
    flat = flatten.flatten_world([smoke, xz_rect(-20, 20, -20, 20, -1, GREEN)])
    sc = gpu_ctx.scene(flat)
    rng = np.random.default_rng(8)
    n = 20000
    o = rng.uniform(-8, 12, (n, 3)).astype(np.float32)
    o[: n // 3] = (np.array([3.0, 3.0, 4.0]) + rng.uniform(-1.5, 1.5, (n // 3, 3))).astype(np.float32)  # inside
    d = rng.normal(size=(n, 3)).astype(np.float32)
    xi = rng.random((n, 1)).astype(np.float32)
    dev = sc.hit_batch(o, d, medium_xi=xi)
    orc = oracle.hit_batch(flat, o, d, medium_xi=xi)
    assert int(orc[:, 11].sum()) > n // 4
    _check_hits(dev, orc, o, d, 1e-4)
    sc.close()


def test_medium_with_a_general_boundary(gpu_ctx, oracle):
    """constant_medium.clj:9-20 takes ANY hittable as its boundary. One sphere and one closed box are
    answered from a single intersection; everything else (here: a list of a box and a sphere that
    overlap, and a list of two spheres under a translate) runs two closest-hit queries over the
    boundary's groups, in the kernel instantiation compiled for such worlds. Hit records with
    scripted medium uniforms, per-sample radiance with the shared random stream, and a render
    against the sum of its samples."""
    from raytracing_amd import abi
    from raytracing_amd.core import build_camera
    from raytracing_amd.hittables import sphere
    from raytracing_amd.materials import diffuse_light
    blob = hittable_list([box([0, 0, 0], [4, 5, 4], WHITE), sphere([4, 3, 2], 2.5, WHITE)])
    pair = translate(hittable_list([sphere([0, 0, 0], 2.0, WHITE), sphere([2.5, 0.5, 0], 1.5, WHITE)]), [-7, 3, 1])
    world = [constant_medium(blob, 0.4, [0.8, 0.3, 0.3]), constant_medium(pair, 0.7, [0.2, 0.2, 0.9]),
             xz_rect(-30, 30, -30, 30, -1, GREEN), xz_rect(-6, 8, -4, 8, 12, diffuse_light([5, 5, 5]))]
    flat = flatten.flatten_world(world)
    sc = gpu_ctx.scene(flat)
    rng = np.random.default_rng(12)
    n = 20000
    o = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    o[: n // 4] = (np.array([3.0, 2.5, 2.0]) + rng.uniform(-2, 2, (n // 4, 3))).astype(np.float32)      # inside the blob
    o[n // 4: n // 2] = (np.array([-6.0, 3.0, 1.0]) + rng.uniform(-2, 2, (n // 4, 3))).astype(np.float32)  # inside the pair
    d = rng.normal(size=(n, 3)).astype(np.float32)
    xi = rng.random((n, 2)).astype(np.float32)
    dev = sc.hit_batch(o, d, medium_xi=xi)
    orc = oracle.hit_batch(flat, o, d, medium_xi=xi)
    assert int((orc[:, 11] > 0.5).sum()) > n // 4
    _check_hits(dev, orc, o, d, 1e-4)
    # the render path: per-sample radiance against the oracle with the shared random stream
    cfg = {"lookfrom": [2, 6, -26], "lookat": [0, 2, 0], "vfov": 40.0, "aperture": 0.0, "background": [0.1, 0.1, 0.2]}
    cam = build_camera(cfg, 1.0)
    W = H = 40
    p = abi.make_params(W, H, 4, 50, cfg["background"], seed=5)
    m = 4000
    px = rng.integers(0, W * H, m).astype(np.int32)
    smp = rng.integers(0, 4, m).astype(np.int32)
    dev_s = sc.sample_batch(cam, p, px, smp)
    orc_s = oracle.sample_batch(flat, cam, p, px, smp)
    rel = np.abs(dev_s - orc_s).max(axis=1) / np.maximum(np.abs(orc_s).max(axis=1), 1e-2)
    assert (rel > 2e-3).mean() <= 5e-3 and np.median(rel) <= 1e-5
    # ... and the render kernel against the per-sample kernel: one sample per chunk, summed in order
    rgb, _, st = sc.render(cam, p)
    allpx = np.repeat(np.arange(W * H, dtype=np.int32), 4)
    alls = np.tile(np.arange(4, dtype=np.int32), W * H)
    per = sc.sample_batch(cam, p, allpx, alls).reshape(W * H, 4, 3)
    acc = np.zeros((W * H, 3), dtype=np.float32)
    for s in range(4):
        acc += per[:, s]
    assert np.array_equal(acc.reshape(H, W, 3), rgb)
    assert st["samples"] == W * H * 4
    sc.close()


def _lattice_world(rng):
    """Rects on a coarse lattice, so that extents coincide by chance: complete boxes, boxes with one, two or
    three faces missing, boxes of several materials, walls shared between neighbours, duplicates; plus blocks
    under chains and a few spheres. The host has to find the (open) boxes among them without ever
    changing an answer."""
    mats = [RED, GREEN, WHITE]
    world = []
    for _ in range(int(rng.integers(2, 6))):
        p0 = rng.integers(-3, 3, 3) * 4.0
        size = rng.integers(1, 4, 3) * 4.0
        p1 = p0 + size
        faces = [xy_rect(p0[0], p1[0], p0[1], p1[1], p0[2], mats[int(rng.integers(3))]),
                 xy_rect(p0[0], p1[0], p0[1], p1[1], p1[2], mats[int(rng.integers(3))]),
                 xz_rect(p0[0], p1[0], p0[2], p1[2], p0[1], mats[int(rng.integers(3))]),
                 xz_rect(p0[0], p1[0], p0[2], p1[2], p1[1], mats[int(rng.integers(3))]),
                 yz_rect(p0[1], p1[1], p0[2], p1[2], p0[0], mats[int(rng.integers(3))]),
                 yz_rect(p0[1], p1[1], p0[2], p1[2], p1[0], mats[int(rng.integers(3))])]
        keep = rng.permutation(6)[: int(rng.integers(3, 7))]
        world += [faces[i] for i in sorted(keep, key=lambda _: rng.random())]
        if rng.random() < 0.3:
            world.append(faces[int(keep[0])])      # a duplicate wall
    for _ in range(int(rng.integers(0, 3))):
        b = box([0, 0, 0], list(rng.integers(1, 4, 3) * 2.0), mats[int(rng.integers(3))])
        world.append(translate(rotate_y(b, float(rng.uniform(-60, 60))), list(rng.uniform(-8, 8, 3))))
    from raytracing_amd.hittables import sphere
    for _ in range(int(rng.integers(0, 3))):
        world.append(sphere(list(rng.uniform(-10, 10, 3)), float(rng.uniform(0.5, 3)), mats[int(rng.integers(3))]))
    order = rng.permutation(len(world))
    return [world[i] for i in order]


# (RT_LATTICE_SWEEP=<first seed>:<count> runs other worlds than the suite's twelve)
_SEEDS = range(12) if not os.environ.get("RT_LATTICE_SWEEP") else range(*(lambda a, b: (a, a + b))(*map(int, os.environ["RT_LATTICE_SWEEP"].split(":"))))


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_lattice_worlds_against_the_oracle(gpu_ctx, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    flat = flatten.flatten_world(_lattice_world(rng))
    sc = gpu_ctx.scene(flat)
    n = 6000
    o = rng.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = (rng.uniform(-10, 10, (n, 3)) - o).astype(np.float32)       # towards the middle, where the walls are
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    assert int(orc[:, 11].sum()) > n // 20
    _check_hits(dev, orc, o, d, 2e-4, requery=lambda oo: oracle.hit_batch(flat, oo, d))
    rv, cover = sc.variants()
    for v in cover:                                    # every instantiation that covers the world agrees bit for bit
        assert sc.hit_batch(o, d, variant=v).tobytes() == dev.tobytes()
    sc.close()


def test_a_rect_that_coincides_with_a_box_face_wins_when_it_comes_later(gpu_ctx, oracle):
    """rectangles.clj accepts t = t-max, so of two coinciding rects the LATER one in the scan wins. The
    device tests loose rects before boxes: a box (or room) whose face coincides with a later loose rect is
    therefore not merged. All four orders, exact rays."""
    room = [xy_rect(0, 4, 0, 4, 4, WHITE), xz_rect(0, 4, 0, 4, 0, WHITE), xz_rect(0, 4, 0, 4, 4, WHITE),
            yz_rect(0, 4, 0, 4, 0, WHITE), yz_rect(0, 4, 0, 4, 4, WHITE)]
    patch = xy_rect(1, 3, 1, 3, 4, GREEN)
    worlds = {"box then rect": [box([0, 0, 0], [4, 4, 4], WHITE), xy_rect(0, 4, 0, 4, 4, GREEN)],
              "rect then box": [xy_rect(0, 4, 0, 4, 4, GREEN), box([0, 0, 0], [4, 4, 4], WHITE)],
              "room then rect": room + [patch], "rect then room": [patch] + room}
    o = np.float32([[2, 2, 2]] * 4)
    d = np.float32([[0, 0, 1], [0.1, 0.05, 1], [0.2, -0.1, 1], [-0.3, 0.2, 1]])
    for name, world in worlds.items():
        flat = flatten.flatten_world(world)
        sc = gpu_ctx.scene(flat)
        dev = sc.hit_batch(o, d)
        orc = oracle.hit_batch(flat, o, d)
        assert dev["hit"].all() and orc[:, 11].all(), name
        assert np.array_equal(dev["mat_id"], orc[:, 10].astype(np.int32)), name
        assert np.array_equal(dev["t"], orc[:, 0].astype(np.float32)), name
        sc.close()
