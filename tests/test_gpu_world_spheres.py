"""Spheres below translate / rotate-y wrappers (base.clj:229-258, :263-302 under translate.clj:8-24 and
rotate.clj:11-47). The device intersects them in WORLD space (a chain is a rigid motion, so t is the same) inside
the groups and trees of the chain-less primitives, and still builds their records in the chain's space and unwinds
them through the chain (Q5: the record side of rotate-y is not the inverse of its ray side). These worlds are the
shape of final-scene's cluster (scenes.clj:127-131,143-145): a tree of spheres below a turn and a shift, beside
chain-less primitives, plus a moving sphere below a chain of its own. Records against the oracle, which walks the
wrappers as the reference does."""
import random

import numpy as np
import pytest

from raytracing_amd import flatten
from raytracing_amd.hittables import bvhnode, hittable_list, moving_sphere, rotate_y, sphere, translate, xz_rect
from raytracing_amd.materials import dielectric, lambertian, metal
from test_gpu_parity import _check_hits

pytestmark = pytest.mark.gpu


def _world(n_cluster, seed):
    rng = random.Random(seed)
    white = lambertian([0.73, 0.73, 0.73])
    cluster = [sphere([rng.uniform(0, 165), rng.uniform(0, 165), rng.uniform(0, 165)], 10, white) for _ in range(n_cluster)]
    loose = [sphere([rng.uniform(-200, 400), rng.uniform(20, 300), rng.uniform(-100, 300)], rng.uniform(15, 50),
                    rng.choice([white, metal([0.8, 0.8, 0.9], 1.0), dielectric(1.5)])) for _ in range(24)]
    return [
        xz_rect(-600, 800, -600, 800, 0, lambertian([0.48, 0.83, 0.53])),
        translate(rotate_y(bvhnode(hittable_list(cluster), 0.0, 1.0, rng), 15), [-100, 270, 395]),
        translate(rotate_y(moving_sphere([0, 0, 0], [30, 0, 0], 0.0, 1.0, 50, lambertian([0.7, 0.3, 0.1])), -40), [400, 400, 200]),
        rotate_y(translate(sphere([0, 0, 0], 35, dielectric(1.5)), [120, 90, -40]), 70),
    ] + loose


@pytest.mark.parametrize("n_cluster,seed", [(40, 1), (300, 2), (1000, 3)])
def test_spheres_below_chains_against_the_oracle(gpu_ctx, oracle, n_cluster, seed):
    flat = flatten.flatten_world(_world(n_cluster, seed))
    sc = gpu_ctx.scene(flat)
    info = sc.info()
    # every sphere stands in the chain-less groups: ONE tree over them all (the plane, 70 times their size, is kept
    # out of it and scanned on its own); a tree per chain would leave fewer nodes
    assert info["n_bvh_nodes"] == info["n_prims_flat"] - 2
    rng = np.random.default_rng(seed)
    n = 12000
    o = np.stack([rng.uniform(-300, 700, n), rng.uniform(5, 600, n), rng.uniform(-500, 700, n)], axis=1).astype(np.float32)
    # half of the rays at the cluster (where it stands in the world), half anywhere
    c, s = np.cos(np.radians(15)), np.sin(np.radians(15))
    local = rng.uniform(0, 165, (n, 3))
    target = np.stack([c * local[:, 0] + s * local[:, 2] - 100, local[:, 1] + 270, -s * local[:, 0] + c * local[:, 2] + 395], axis=1)
    target[n // 2:] = np.stack([rng.uniform(-200, 500, n - n // 2), rng.uniform(0, 450, n - n // 2), rng.uniform(-200, 500, n - n // 2)], axis=1)
    d = (target - o).astype(np.float32)
    tm = rng.random(n).astype(np.float32)
    dev = sc.hit_batch(o, d, tm)
    orc = oracle.hit_batch(flat, o, d, tm)
    assert (orc[:n // 2, 11] > 0.5).mean() > 0.25  # the rays at the cluster do meet something
    _check_hits(dev, orc, o, d, 5e-5, label=f"spheres below chains ({n_cluster} in the tree)",
                requery=lambda oo: oracle.hit_batch(flat, oo, d, tm))
    # every instantiation that covers the world answers the same, bit for bit
    _, covering = sc.variants()
    for v in covering:
        again = sc.hit_batch(o, d, tm, variant=v)
        assert dev.tobytes() == again.tobytes(), v
    sc.close()
