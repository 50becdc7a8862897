"""Where a hit's distance error comes from, world by world: for random worlds of tests/test_gpu_fuzz.py, the records
with the largest |t - t_ref| |d| in ulps of the largest coordinate, with the cosine of incidence (from the oracle's
normal) and the product of the two. Usage (GPU box): python tools/dist_probe.py profile:seed:n_objects ...
or --sweep first:count for the maxima over a range of the sweep's worlds."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracing_amd import abi, flatten, lib  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from oracle import binding as oracle  # noqa: E402

ctx = lib.Context(0)


def batch(profile, seed, n_objects):
    rng = np.random.default_rng(7000 + seed)
    extent = 6.0 + 0.15 * min(n_objects, 100)
    flat = flatten.flatten_world(F._world(rng, n_objects, extent, F.PROFILES[profile]))
    sc = ctx.scene(flat)
    nm = flat.desc.n_media
    n = 8000
    o = rng.uniform(-1.6 * extent, 1.6 * extent, (n, 3)).astype(np.float32)
    d = (rng.uniform(-extent, extent, (n, 3)) - o).astype(np.float32)
    d[: n // 4] = rng.normal(size=(n // 4, 3)).astype(np.float32)
    tm = rng.random(n).astype(np.float32)
    xi = rng.random((n, nm)).astype(np.float32) if nm else None
    dev = sc.hit_batch(o, d, tm, medium_xi=xi)
    orc = oracle.hit_batch(flat, o, d, tm, medium_xi=xi)
    return flat, o, d, dev, orc


def measures(flat, o, d, dev, orc):
    both = dev["hit"].astype(bool) & (orc[:, 11] > 0.5) & (dev["mat_id"] == orc[:, 10].astype(np.int32))
    dn = np.linalg.norm(d.astype(np.float64), axis=1)
    dist = np.abs(dev["t"].astype(np.float64) - orc[:, 0]) * dn
    size = np.maximum(np.maximum(np.abs(o).max(axis=1), np.abs(orc[:, 1:4]).max(axis=1)), 1.0)
    ulps = dist / (size * 2.0 ** -24)
    nrm = orc[:, 4:7]
    cos = np.abs((nrm * d).sum(axis=1)) / np.maximum(np.linalg.norm(nrm, axis=1) * dn, 1e-300)
    mtype = np.asarray(flat.a_mat_type)[np.clip(orc[:, 10].astype(int), 0, len(flat.a_mat_type) - 1)] if hasattr(flat, "a_mat_type") else np.zeros(len(o), int)
    return both, ulps, cos, mtype


if sys.argv[1] == "--sweep":
    first, count = (int(v) for v in sys.argv[2].split(":"))
    sizes = [4, 6, 9, 12, 14, 20, 30, 45, 5, 8, 60, 16, 25, 7, 90, 11, 3, 200, 2, 35]
    names = list(F.PROFILES)
    worst = []
    for i in range(count):
        s = first + i
        case = [c for c in F.CASES if c[1] == s]
        if not case:
            continue
        flat, o, d, dev, orc = batch(*case[0])
        both, ulps, cos, mtype = measures(flat, o, d, dev, orc)
        if both.any():
            k = np.argmax(np.where(both, ulps, 0)); k2 = np.argmax(np.where(both, ulps * cos, 0))
            worst.append((ulps[k], cos[k], ulps[k2] * cos[k2], cos[k2], case[0]))
    worst.sort(key=lambda w: -w[0])
    print("largest distance errors (ulps, cosine there):")
    for w in worst[:12]:
        print(f"  {w[0]:8.1f} ulps at cos {w[1]:.4f}   {w[4]}")
    worst.sort(key=lambda w: -w[2])
    print("largest distance error x cosine of incidence:")
    for w in worst[:12]:
        print(f"  {w[2]:8.1f} at cos {w[3]:.4f}   {w[4]}")
else:
    for a in sys.argv[1:]:
        p, s, n = a.split(":")
        flat, o, d, dev, orc = batch(p, int(s), int(n))
        both, ulps, cos, mtype = measures(flat, o, d, dev, orc)
        idx = np.argsort(-np.where(both, ulps, 0))[:8]
        print(a)
        for k in idx:
            print(f"  {ulps[k]:8.1f} ulps  cos {cos[k]:.5f}  x cos {ulps[k] * cos[k]:7.2f}  t {orc[k, 0]:.6g}  |o| {np.abs(o[k]).max():.3g}  "
                  f"mat {int(orc[k, 10])}  normal {orc[k, 4:7]}")
