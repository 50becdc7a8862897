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
    device and the oracle): Philox-4x32-7, counter (px, smp, m>>2, bounce<<4|4), key (seed lo, seed hi)."""
    from oracle import binding as ob
    o = ob.philox([px, smp, m >> 2, (bounce << 4) | 4], [seed & 0xFFFFFFFF, seed >> 32])
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


# ------------------------------------------------------------------ independent re-derivations
_MASK = 0xFFFFFFFF


def philox4x32(ctr, key, rounds=7):
    """Philox-4x32 (Salmon et al., SC'11) in plain Python, independent of both the oracle and the
    device; pinned by the Random123 known-answer vectors in tests/test_gpu_kat.py."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(rounds):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> 32) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0, k1 = (k0 + 0x9E3779B9) & _MASK, (k1 + 0xBB67AE85) & _MASK
    return [c0, c1, c2, c3]


def keyed_camera_ray(cam, W, H, px, smp, seed):
    """core.clj:96-102 + ray.clj:14-26 in FP64 for pixel index px, sample smp, lens radius 0: the
    jitter uniforms are words 0..2 of the keyed block (counter (px, smp, 0, 0), key (seed lo,
    seed hi)), top 24 bits each. -> (orig[3], dir[3], time)."""
    w = philox4x32([px, smp, 0, 0], [seed & _MASK, seed >> 32])
    u = [(x >> 8) / 16777216.0 for x in w]
    x, y = px % W, px // W
    s, t = (x + u[0]) / (W - 1), (y + u[1]) / (H - 1)
    org = np.array(cam.origin[:], dtype=np.float64)
    d = (np.array(cam.lower_left_corner[:]) + s * np.array(cam.horizontal[:]) + t * np.array(cam.vertical[:])) - org
    return org, d, cam.time0 + (cam.time1 - cam.time0) * u[2]


def cornell_block_walk(o, d):
    """Cornell block 1 (scenes.clj:82-84: translate [265 0 295] of rotate-y 15 of box [0 0 0]-[165 330 165])
    hit by ray (o, d), walked by hand in FP64 numpy: translate.clj:10-17, rotate.clj:11-47, box.clj:18-23
    order, base.clj:92-101 scan. -> (t, p, normal, front_face) or None."""
    import math
    s, c = math.sin(math.radians(15)), math.cos(math.radians(15))
    om = o - np.array([265.0, 0, 295.0])
    orot = np.array([c * om[0] - s * om[2], om[1], s * om[0] + c * om[2]])
    drot = np.array([c * d[0] - s * d[2], d[1], s * d[0] + c * d[2]])
    best = None
    sides = [(2, 165.0, (0, 1), (0, 165, 0, 330)), (2, 0.0, (0, 1), (0, 165, 0, 330)),
             (1, 330.0, (0, 2), (0, 165, 0, 165)), (1, 0.0, (0, 2), (0, 165, 0, 165)),
             (0, 165.0, (1, 2), (0, 330, 0, 165)), (0, 0.0, (1, 2), (0, 330, 0, 165))]
    for axis, k, (ia, ib), (a0, a1, b0, b1) in sides:
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (k - orot[axis]) / drot[axis]
        if t < 0.001 or (best is not None and t > best[0]):
            continue
        a, b = orot[ia] + t * drot[ia], orot[ib] + t * drot[ib]
        if a < a0 or a > a1 or b < b0 or b > b1:
            continue
        n = np.zeros(3)
        n[axis] = 1.0
        if drot @ n >= 0:
            n = -n
        best = (t, orot + t * drot, n)
    if best is None:
        return None
    t, p, n = best
    p2 = np.array([c * p[0] + s * p[2], p[1], s * p[0] + c * p[2]])   # rotate.clj:32-36, +sin in z
    n2 = np.array([c * n[0] + s * n[2], n[1], s * n[0] + c * n[2]])
    if not (drot @ n2 < 0):                                            # rotate.clj:43
        n2 = -n2
    front = (d @ n2) < 0                                               # translate.clj:13
    if not front:
        n2 = -n2
    return t, p2 + np.array([265.0, 0, 295.0]), n2, bool(front)
