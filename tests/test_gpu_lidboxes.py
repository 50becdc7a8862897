"""Boxes with a sunk lid as single tree leaves. box.clj:16-23 builds a box as six rects, and the ^long field k of
rectangles.clj:9,46,83 drops every rect's plane to a whole number: a box on y0 = 0 with its top at 37.4 has walls up to
37.4 and its lid at 37, a rim of 0.4 around a recess (the 400 ground boxes of final-scene, scenes.clj:110-120). The
device makes ONE leaf of such a box (floor and walls as an open box, the lid as its rect); the oracle scans the six
rects as the reference does. The rays go where the two could part: down into the recess, across rims, out of the
recess (origins on the lid), along shared walls between neighbours of different heights, along the axes."""
import numpy as np
import pytest

from raytracing_amd import flatten
from raytracing_amd.hittables import box, bvhnode, hittable_list, sphere, xz_rect
from raytracing_amd.materials import diffuse_light, lambertian, metal
from test_gpu_parity import _check_hits

pytestmark = pytest.mark.gpu


def _ground(n_side, seed, w=10.0, two_tone=False):
    """n_side x n_side boxes of width w on y = 0 with random tops (some whole numbers: lid on the rim), as the
    ground of final-scene; below a light and beside two spheres."""
    import random
    rng = random.Random(seed)
    ground, other = lambertian([0.48, 0.83, 0.53]), lambertian([0.8, 0.3, 0.3])
    boxes = []
    for i in range(n_side):
        for j in range(n_side):
            x0, z0 = -0.5 * n_side * w + i * w, -0.5 * n_side * w + j * w
            y1 = float(rng.randint(1, 9)) if rng.random() < 0.15 else rng.uniform(1.0, 10.1)
            # (two materials: the boxes stand apart, a wall shared by two materials would tie exactly and the
            # reference settles that tie by the shape of its own tree)
            x1, z1 = (x0 + w - 1, z0 + w - 1) if two_tone else (x0 + w, z0 + w)
            boxes.append(box([x0, 0.0, z0], [x1, y1, z1], other if two_tone and (i + j) % 3 == 0 else ground))
    return [bvhnode(hittable_list(boxes), 0.0, 1.0, rng),
            xz_rect(-20, 20, -20, 20, 30, diffuse_light([7, 7, 7])),
            sphere([0, 15, 0], 4, metal([0.8, 0.8, 0.9], 0.0)),
            sphere([12, 14, -7], 3, lambertian([0.2, 0.4, 0.9]))]


def _rays(rng, n, n_side, w=10.0):
    half = 0.5 * n_side * w
    o = np.stack([rng.uniform(-half - 10, half + 10, n), rng.uniform(0.5, 40, n), rng.uniform(-half - 10, half + 10, n)], axis=1)
    d = rng.normal(size=(n, 3))
    k = n // 5
    d[:k, 1] = -np.abs(d[:k, 1]) * 4                                   # steeply down, into the recesses
    o[k:2 * k, 1] = rng.uniform(1.0, 10.2, k)                          # low: across rims and along walls
    d[k:2 * k, 1] *= 0.05
    o[2 * k:3 * k, 1] = np.floor(rng.uniform(1.0, 10.0, k))            # from the height of a lid, upwards or level
    d[2 * k:3 * k, 1] = np.abs(d[2 * k:3 * k, 1]) * 0.3
    axis = rng.integers(0, 3, k)                                       # exactly along an axis
    d[3 * k:4 * k] = 0
    d[np.arange(3 * k, 4 * k), axis] = rng.choice([-1.0, 1.0], k)
    o[3 * k:4 * k, 1] = rng.uniform(0.5, 12, k)
    return o.astype(np.float32), d.astype(np.float32)


@pytest.mark.parametrize("n_side,seed,two_tone", [(5, 1, False), (12, 2, False), (20, 3, False), (12, 4, True)])
def test_boxes_with_sunk_lids_against_the_rect_by_rect_oracle(gpu_ctx, oracle, n_side, seed, two_tone):
    flat = flatten.flatten_world(_ground(n_side, seed, two_tone=two_tone))
    sc = gpu_ctx.scene(flat)
    info = sc.info()
    # one leaf per box: far fewer nodes than six rects per box would make
    assert 0 < info["n_bvh_nodes"] < 2 * n_side * n_side + 8, info
    o, d = _rays(np.random.default_rng(seed), 40000, n_side)
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    assert int(orc[:, 11].sum()) > 15000
    # neighbours share walls: on such a wall two rects tie exactly and u, v name different rects (the solid colours never read them)
    ground_mats = {int(m) for m in np.unique(orc[orc[:, 11] > 0.5, 10])}
    _check_hits(dev, orc, o, d, 1e-4, uv_exempt=tuple(ground_mats), label=f"{n_side * n_side} boxes with sunk lids")
    _, covering = sc.variants()
    for v in covering:
        assert sc.hit_batch(o, d, variant=v).tobytes() == dev.tobytes(), v
    sc.close()


def test_a_box_whose_rect_shares_a_plane_with_a_loose_rect_keeps_its_six_leaves(gpu_ctx, oracle):
    # the patch lies in the plane of the lids of height-3 boxes... every box here has its lid at y = 3: all are given up
    import random
    rng = random.Random(9)
    g = lambertian([0.5, 0.5, 0.5])
    boxes = [box([10.0 * i, 0.0, 10.0 * j], [10.0 * i + 10, 3.5, 10.0 * j + 10], g) for i in range(4) for j in range(4)]
    world = [bvhnode(hittable_list(boxes), 0.0, 1.0, rng), xz_rect(-5, 45, -5, 45, 3, lambertian([0.9, 0.1, 0.1]))]
    flat = flatten.flatten_world(world)
    sc = gpu_ctx.scene(flat)
    assert sc.info()["n_bvh_nodes"] >= 6 * 16 - 1  # no box leaf: the patch must win or lose its ties by the scan
    nrng = np.random.default_rng(9)
    o = np.stack([nrng.uniform(-5, 45, 20000), nrng.uniform(3.1, 20, 20000), nrng.uniform(-5, 45, 20000)], axis=1).astype(np.float32)
    d = nrng.normal(size=(20000, 3)).astype(np.float32)
    d[:, 1] = -np.abs(d[:, 1])
    dev = sc.hit_batch(o, d)
    orc = oracle.hit_batch(flat, o, d)
    hit = orc[:, 11] > 0.5
    assert np.array_equal(dev["hit"] != 0, hit)
    # lid and patch coincide over every box: the material names the winner of the tie
    assert np.array_equal(dev["mat_id"][hit], orc[hit, 10].astype(np.int32))
    sc.close()
