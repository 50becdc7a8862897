"""Worlds far larger than the reference's (50 000 to 1.2 million spheres in a jittered grid above a ground): scene creation
time per tree builder, tree height and LDS need, closest-hit records of 4 096 rays compared between builders (they must be
bit-identical: the closest hit does not depend on the tree), and a small render. Beyond 2^20 primitives per group the
default builder is the Morton / radix-tree one. Usage (GPU box): python tools/scale_probe.py [n_spheres ...]"""
import sys
import time

import numpy as np

from raytracing_amd import abi, flatten, lib
from raytracing_amd.camera import camera
from raytracing_amd.hittables import sphere
from raytracing_amd.materials import dielectric, lambertian, metal

sizes = [int(a) for a in sys.argv[1:]] or [50_000, 300_000]
ctx = lib.Context(0)
rng = np.random.default_rng(5)
mats = [lambertian([0.7, 0.3, 0.2]), lambertian([0.2, 0.5, 0.7]), metal([0.8, 0.8, 0.7], 0.1), dielectric(1.5)]
for n in sizes:
    side = int(np.ceil(np.sqrt(n)))
    t0 = time.time()
    gx, gz = np.meshgrid(np.arange(side), np.arange(side))
    c = np.stack([gx.ravel()[:n] + rng.uniform(0, 0.6, n), np.full(n, 0.2), gz.ravel()[:n] + rng.uniform(0, 0.6, n)], axis=1)
    world = [sphere([0.0, -1000.0, 0.0], 1000.0, mats[0])]
    world += [sphere([float(x), float(y), float(z)], 0.2, mats[i & 3]) for i, (x, y, z) in enumerate(c)]
    flat = flatten.flatten_world(world)
    t_flat = time.time() - t0
    o = np.stack([rng.uniform(0, side, 4096), rng.uniform(0.5, 30, 4096), rng.uniform(0, side, 4096)], axis=1).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    d[:, 1] = -np.abs(d[:, 1]) - 0.05
    tm = np.zeros(4096, np.float32)
    recs = {}
    builders = [("default", abi.RT_BVH_DEFAULT), ("lbvh", abi.RT_BVH_DEVICE_LBVH)]
    if n <= 32768:
        builders.append(("host-sah", abi.RT_BVH_HOST_SAH))
    for name, b in builders:
        t0 = time.time()
        sc = ctx.scene(flat, bvh_builder=b)
        dt = time.time() - t0
        info = sc.info()
        digest, n_nodes, height = sc.bvh_digest()
        t0 = time.time()
        recs[name] = np.array(sc.hit_batch(o, d, tm))
        t_hit = time.time() - t0
        cam = abi.make_camera(camera([side * 0.5, 8.0, -6.0], [side * 0.5, 0.0, side * 0.3], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0))
        p = abi.make_params(256, 256, 4, 50, [0.7, 0.8, 1.0], seed=1)
        st = sc.render_device(cam, p, 0, 0, want_stats=True)
        print(f"n={n:8d} {name:8s}: create {dt * 1e3:8.1f} ms  nodes {n_nodes} height {height} lds {info['lds_bytes']} B  "
              f"hit_batch {t_hit * 1e3:6.1f} ms  256x256@4 kernel {st['kernel_ms']:7.2f} ms ({st['rays'] / st['samples']:.2f} rays/sample)", flush=True)
        sc.close()
    names = list(recs)
    for other in names[1:]:
        same = recs[names[0]].tobytes() == recs[other].tobytes()
        print(f"           records {names[0]} == {other}: {same}; hits {int((recs[names[0]]['hit'] != 0).sum())} of 4096", flush=True)
        assert same
    print(f"           (python flattening of the world: {t_flat:.1f} s)", flush=True)
ctx.close()
