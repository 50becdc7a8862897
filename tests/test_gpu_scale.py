"""A world far larger than any of the reference's: 60 000 spheres in a jittered grid above a ground sphere. The closest hit
does not depend on the tree, so the records of the GPU-built surface-area tree and of the Morton tree must be equal byte
for byte, the trees must differ, and a render must trace what the scene's size suggests (tools/scale_probe.py runs the
same up to 1.2 million spheres)."""
import numpy as np
import pytest

from raytracing_amd import abi, flatten
from raytracing_amd.camera import camera
from raytracing_amd.hittables import sphere
from raytracing_amd.materials import dielectric, lambertian, metal

pytestmark = pytest.mark.gpu


def test_sixty_thousand_spheres_on_both_device_builders(gpu_ctx):
    rng = np.random.default_rng(11)
    n, side = 60_000, 245
    gx, gz = np.meshgrid(np.arange(side), np.arange(side))
    c = np.stack([gx.ravel()[:n] + rng.uniform(0, 0.6, n), np.full(n, 0.2), gz.ravel()[:n] + rng.uniform(0, 0.6, n)], axis=1)
    mats = [lambertian([0.7, 0.3, 0.2]), lambertian([0.2, 0.5, 0.7]), metal([0.8, 0.8, 0.7], 0.1), dielectric(1.5)]
    world = [sphere([0.0, -1000.0, 0.0], 1000.0, mats[0])]
    world += [sphere([float(x), float(y), float(z)], 0.2, mats[i & 3]) for i, (x, y, z) in enumerate(c)]
    flat = flatten.flatten_world(world)
    o = np.stack([rng.uniform(0, side, 4096), rng.uniform(0.5, 30, 4096), rng.uniform(0, side, 4096)], axis=1).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d[:, 1] = -np.abs(d[:, 1]) - 0.05
    got = {}
    for builder in (abi.RT_BVH_DEFAULT, abi.RT_BVH_DEVICE_LBVH):
        sc = gpu_ctx.scene(flat, bvh_builder=builder)
        digest, n_nodes, height = sc.bvh_digest()
        assert n_nodes == n - 1 and 14 <= height <= 40      # (the ground is kept out of the tree: far larger than the rest)
        recs = sc.hit_batch(o, d, np.zeros(4096, np.float32))
        cam = abi.make_camera(camera([side * 0.5, 8.0, -6.0], [side * 0.5, 0.0, side * 0.3], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0))
        st = sc.render_device(cam, abi.make_params(128, 128, 4, 50, [0.7, 0.8, 1.0], seed=1), 0, 0, want_stats=True)
        assert st["samples"] == 128 * 128 * 4 and st["rays"] > st["samples"]
        got[builder] = (digest, np.array(recs).tobytes(), st["rays"], int((np.array(recs)["hit"] != 0).sum()))
        sc.close()
    a, b = got[abi.RT_BVH_DEFAULT], got[abi.RT_BVH_DEVICE_LBVH]
    assert a[0] != b[0], "the two builders gave the same tree: one of them did not run"
    assert a[1] == b[1], "closest-hit records differ between the tree builders"
    assert a[2] == b[2], "the renders traced different numbers of rays"
    assert a[3] > 2000
