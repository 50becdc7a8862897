"""Generates tests/golden/*.npz from the CPU oracle.

These are REGRESSION fixtures of this repository's own oracle (the reference ships no golden
vectors and cannot be run: no JVM). They freeze the oracle's answers for fixed inputs so that a
later change to the oracle, the scene mirrors, the flattener or the device kernels that moves a
result shows up as a diff. Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import binding as ob  # noqa: E402
from raytracing_amd import abi  # noqa: E402
from util import ALL_SCENES, medium_uniform, probe_rays, scene_bundle  # noqa: E402

SEED = 0x5EED0001


def main():
    for name in ALL_SCENES:
        cfg, flat, cam = scene_bundle(name)
        o, d, tm = probe_rays(cfg, 256, seed=1)
        nm = flat.desc.n_media
        xi = np.random.default_rng(2).random((256, max(nm, 1))).astype(np.float32) if nm else None
        hits = ob.hit_batch(flat, o, d, tm, medium_xi=xi)
        p = abi.make_params(16, 16, 4, 50, cfg["background"], seed=SEED)
        rgb, argb, rays = ob.render(flat, cam, p, nthreads=4)
        out = os.path.join(HERE, name + ".npz")
        np.savez_compressed(out, orig=o, dir=d, time=tm, xi=xi if xi is not None else np.zeros((0, 0), np.float32),
                            hits=hits, rgb_sum=rgb, argb=argb, rays=np.int64(rays),
                            n_prims=np.int32(flat.desc.n_prims), n_nodes=np.int32(flat.desc.n_nodes))
        print(name, "hits", int(hits[:, 11].sum()), "rays", rays, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
