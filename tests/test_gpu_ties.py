"""Exact ties between primitives of different instance chains. Two XZ rects at one height under different rotate-y
wrappers tie exactly for every ray through their overlap (the turn leaves y alone; rotate.clj:18-27), and the
reference's scan (base.clj:92-101 with rectangles.clj's `t <= t-max`) gives the hit to the one that comes LATER in
the world vector. The device tests a world group by group, so its groups have to follow the scan."""
import itertools

import numpy as np
import pytest

from raytracing_amd import flatten
from raytracing_amd.hittables import rotate_y, translate, xy_rect, xz_rect
from raytracing_amd.materials import lambertian

pytestmark = pytest.mark.gpu


def _patch(i, angle, wrapped=True):
    r = xz_rect(-3 + 0.1 * i, 3, -3, 3 - 0.1 * i, 1.0, lambertian([0.1 * (i + 1), 0.5, 0.5]))
    return rotate_y(r, angle) if wrapped else r


@pytest.mark.parametrize("order", list(itertools.permutations(range(4)))[::3])
def test_coplanar_rects_under_different_wrappers_fall_as_in_the_scan(gpu_ctx, oracle, order):
    # 0 and 2 lie in the world itself (one chain, not adjacent in the scan), 1 and 3 below two different turns
    patches = [_patch(0, 0.0, False), _patch(1, 20.0), _patch(2, 0.0, False), _patch(3, -35.0)]
    world = [patches[i] for i in order] + [xy_rect(-5, 5, -5, 5, 9, lambertian([0.5, 0.5, 0.5]))]
    flat = flatten.flatten_world(world)
    sc = gpu_ctx.scene(flat)
    rng = np.random.default_rng(sum(k * 7 ** i for i, k in enumerate(order)))
    n = 4000
    o = np.stack([rng.uniform(-4, 4, n), rng.uniform(3, 8, n) * rng.choice([-1, 1], n), rng.uniform(-4, 4, n)], axis=1).astype(np.float32)
    target = np.stack([rng.uniform(-3.4, 3.4, n), np.ones(n), rng.uniform(-3.4, 3.4, n)], axis=1)
    d = (target - o).astype(np.float32)
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    hit = orc[:, 11] != 0
    assert hit.sum() > n // 2
    assert np.array_equal(dev["hit"] != 0, hit)
    # the material names the winner: all four patches overlap around the origin (four-way ties), fewer towards the rim
    assert np.array_equal(dev["mat_id"][hit], orc[hit, 10].astype(np.int32))
    assert len(np.unique(orc[hit, 10])) >= 2
    sc.close()


def test_a_chain_met_twice_is_not_pulled_ahead_of_what_lies_between(gpu_ctx, oracle):
    # translate-only chains tie exactly when the numbers are exact in binary: heights 1 = 0.5 + 0.5 = 2 - 1
    a = xz_rect(-3, 3, -3, 3, 1.0, lambertian([0.1, 0.1, 0.1]))
    b = translate(xz_rect(-3, 3, -3, 3, 2.0, lambertian([0.2, 0.2, 0.2])), [0.0, -1.0, 0.0])
    c = xz_rect(-2, 2, -2, 2, 1.0, lambertian([0.3, 0.3, 0.3]))
    e = translate(xz_rect(-3, 3, -3, 3, 3.0, lambertian([0.4, 0.4, 0.4])), [0.0, -2.0, 0.0])
    for world in ([a, b, c, e], [e, c, b, a], [b, a, e, c], [c, e, a, b]):
        flat = flatten.flatten_world(world)
        sc = gpu_ctx.scene(flat)
        rng = np.random.default_rng(3)
        n = 3000
        # rays on a lattice of eighths: o.y - offset is then exact in FP32 as it is in FP64, and the tie stays exact
        o = (np.stack([rng.integers(-32, 33, n), rng.integers(24, 65, n) * rng.choice([-1, 1], n), rng.integers(-32, 33, n)], axis=1) / 8.0).astype(np.float32)
        d = (np.stack([rng.integers(-20, 21, n) / 8.0, np.ones(n), rng.integers(-20, 21, n) / 8.0], axis=1) - o).astype(np.float32)
        dev = sc.hit_batch(o, d)
        orc = oracle.hit_batch(flat, o, d)
        hit = orc[:, 11] != 0
        assert np.array_equal(dev["hit"] != 0, hit) and hit.sum() > n // 2
        assert np.array_equal(dev["mat_id"][hit], orc[hit, 10].astype(np.int32))
        sc.close()


@pytest.mark.parametrize("padding", [0, 30])
def test_a_wall_listed_twice_wins_from_its_second_place(gpu_ctx, oracle, padding):
    """The scan tests an object as often as the world lists it, and a rect accepts t = closest-so-far: A, B, A with
    A and B in one plane gives the overlap to A (its second test comes after B's), A, B gives it to B. With
    `padding` rects elsewhere the group is big enough for a tree, whose leaf test decides ties on scan ranks."""
    from raytracing_amd.hittables import yz_rect
    A = yz_rect(-4, 4, 4, 8, 8.0, lambertian([0.9, 0.1, 0.1]))
    B = yz_rect(-8, 0, 4, 16, 8.0, lambertian([0.1, 0.9, 0.1]))
    far = [xy_rect(40 + 3 * i, 42 + 3 * i, 40, 42, 50.0 + i, lambertian([0.5, 0.5, 0.5])) for i in range(padding)]
    rng = np.random.default_rng(9)
    n = 3000
    o = np.stack([rng.uniform(10, 14, n) * rng.choice([-1, 1], n) + 8, rng.uniform(-9, 5, n), rng.uniform(3, 17, n)], axis=1).astype(np.float32)
    target = np.stack([np.full(n, 8.0), rng.uniform(-9, 5, n), rng.uniform(3, 17, n)], axis=1)
    d = (target - o).astype(np.float32)
    winners = {}
    for name, world in (("ABA", [A, B, A]), ("AB", [A, B]), ("BAB", [B, A, B]), ("far A far B far A", far[:10] + [A] + far[10:20] + [B] + far[20:] + [A])):
        flat = flatten.flatten_world(world + (far if len(world) <= 3 else []))
        sc = gpu_ctx.scene(flat)
        assert (sc.info()["n_bvh_nodes"] > 0) == (padding > 0)
        dev = sc.hit_batch(o, d)
        orc = oracle.hit_batch(flat, o, d)
        hit = orc[:, 11] != 0
        assert np.array_equal(dev["hit"] != 0, hit) and hit.sum() > n // 3
        assert np.array_equal(dev["mat_id"][hit], orc[hit, 10].astype(np.int32)), name
        both = hit & (target[:, 1] > -4) & (target[:, 1] < 0) & (target[:, 2] > 4) & (target[:, 2] < 8)   # the overlap
        winners[name] = set(int(m) for m in np.unique(dev["mat_id"][both]))
        assert len(winners[name]) == 1 and both.sum() > 50
        sc.close()
    # (materials are numbered as the world first lists them: in "ABA" and "AB" A's is 0, in "BAB" B's is)
    assert winners["ABA"] == {0} and winners["AB"] == {1} and winners["BAB"] == {0}
