"""Shared helpers for the parity tests."""
import numpy as np

from raytracing_amd import abi
from raytracing_amd.core import build_camera, setup

ALL_SCENES = ["random-scene", "two-spheres", "two-perlin-spheres", "earth", "simple-light",
              "cornell-box", "cornell-smoke", "final-scene"]


def scene_bundle(name, scene_seed=3, aspect=1.0):
    cfg, flat = setup(name, scene_seed)
    return cfg, flat, build_camera(cfg, aspect)


def probe_rays(cfg, n, seed):
    """Half camera-like rays towards the look-at region, half rays from random interior points."""
    rng = np.random.default_rng(seed)
    lf = np.array(cfg["lookfrom"], dtype=np.float64)
    la = np.array(cfg["lookat"], dtype=np.float64)
    ext = max(float(np.linalg.norm(lf - la)), 1.0)
    o = np.empty((n, 3))
    d = np.empty((n, 3))
    half = n // 2
    o[:half] = lf
    d[:half] = la + rng.normal(size=(half, 3)) * ext * 0.3 - lf
    o[half:] = la + rng.normal(size=(n - half, 3)) * ext * 0.4
    d[half:] = rng.normal(size=(n - half, 3))
    tm = rng.random(n)
    return o.astype(np.float32), d.astype(np.float32), tm.astype(np.float32)


def medium_uniform(seed, px, smp, bounce, m):
    """The keyed uniform of medium m at ray-color invocation `bounce` (same addressing as the
    device and the oracle): Philox-4x32-7, counter (m>>2, bounce<<4|4, seed), key (px, smp)."""
    from oracle import binding as ob
    o = ob.philox([m >> 2, (bounce << 4) | 4, seed & 0xFFFFFFFF, seed >> 32], [px, smp])
    return (o[m & 3] >> 8) / 16777216.0


def compare_hits(dev, orc, d32):
    """Per-ray error measures between device rt_hit records and oracle rows (see oracle.binding.HIT_FIELDS).

    t: relative to max(|t|, |orig|/|dir| * 2^-4): a ray that starts within rounding distance of a
       surface cannot resolve t relative to t itself once its origin has been rotated in FP32.
    p: Euclidean error relative to max(|p|, 1).  normal: error relative to max(|n|, 1)
       (normals of second-root sphere hits are not unit). u, v: absolute (they live in [0, 1]).
    """
    hd = dev["hit"].astype(bool)
    ho = orc[:, 11] > 0.5
    both = hd & ho
    e = {}
    e["hit_mismatch"] = int((hd != ho).sum())
    e["t"] = np.abs(dev["t"] - orc[:, 0]) / np.maximum(np.abs(orc[:, 0]), 1e-30)
    e["p"] = np.linalg.norm(dev["p"] - orc[:, 1:4], axis=1) / np.maximum(np.linalg.norm(orc[:, 1:4], axis=1), 1.0)
    e["n"] = np.linalg.norm(dev["normal"] - orc[:, 4:7], axis=1) / np.maximum(np.linalg.norm(orc[:, 4:7], axis=1), 1.0)
    du = np.abs(dev["u"] - orc[:, 7])
    dv = np.abs(dev["v"] - orc[:, 8])
    both_nan_u = np.isnan(dev["u"]) & np.isnan(orc[:, 7])
    both_nan_v = np.isnan(dev["v"]) & np.isnan(orc[:, 8])
    e["uv"] = np.maximum(np.where(both_nan_u, 0.0, du), np.where(both_nan_v, 0.0, dv))
    e["front"] = (dev["front_face"] != orc[:, 9].astype(np.int32)) & both
    e["mat"] = (dev["mat_id"] != orc[:, 10].astype(np.int32)) & both
    e["both"] = both
    return e
