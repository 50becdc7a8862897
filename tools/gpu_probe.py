"""Ad-hoc GPU check: hit-record and per-sample parity against the oracle on every scene."""
import random, sys, time
import numpy as np
from raytracing_amd import abi, lib, scenes, flatten
from raytracing_amd.core import build_camera, setup
from oracle import binding as ob

def rays_for(cfg, n, rng):
    lf = np.array(cfg["lookfrom"], dtype=np.float64); la = np.array(cfg["lookat"], dtype=np.float64)
    ext = max(np.linalg.norm(lf - la), 1.0)
    o = np.empty((n, 3)); d = np.empty((n, 3))
    half = n // 2
    # camera-like rays
    o[:half] = lf
    tgt = la + rng.normal(size=(half, 3)) * ext * 0.3
    d[:half] = tgt - lf
    # rays from random interior points
    o[half:] = la + rng.normal(size=(n - half, 3)) * ext * 0.4
    d[half:] = rng.normal(size=(n - half, 3))
    return o.astype(np.float32), d.astype(np.float32)

def cmp_hits(dev, orc):
    hit_d = dev["hit"].astype(bool); hit_o = orc[:, 11] > 0.5
    mism = int((hit_d != hit_o).sum())
    both = hit_d & hit_o
    t_o = orc[both, 0]; t_d = dev["t"][both]
    rel_t = np.abs(t_d - t_o) / np.maximum(np.abs(t_o), 1e-3)
    p_o = orc[both, 1:4]; p_d = dev["p"][both]
    rel_p = np.linalg.norm(p_d - p_o, axis=1) / np.maximum(np.linalg.norm(p_o, axis=1), 1.0)
    n_o = orc[both, 4:7]; n_d = dev["normal"][both]
    err_n = np.linalg.norm(n_d - n_o, axis=1)
    err_uv = np.maximum(np.abs(dev["u"][both] - orc[both, 7]), np.abs(dev["v"][both] - orc[both, 8]))
    ff = int((dev["front_face"][both] != orc[both, 9].astype(int)).sum())
    mat = int((dev["mat_id"][both] != orc[both, 10].astype(int)).sum())
    return dict(n=len(dev), hits=int(both.sum()), hit_mismatch=mism, rel_t=float(np.nanmax(rel_t)) if both.any() else 0,
                rel_p=float(np.nanmax(rel_p)) if both.any() else 0, err_n=float(np.nanmax(err_n)) if both.any() else 0,
                err_uv=float(np.nanmax(err_uv)) if both.any() else 0, front_mismatch=ff, mat_mismatch=mat)

def main():
    names = sys.argv[1:] or list(scenes.SCENES)
    ctx = lib.Context(0)
    rng = np.random.default_rng(7)
    for name in names:
        cfg, flat = setup(name, 3)
        t0 = time.time(); sc = ctx.scene(flat); t_create = time.time() - t0
        print(f"== {name}: {sc.info()} create {t_create*1e3:.1f} ms", flush=True)
        n = 20000
        o, d = rays_for(cfg, n, rng)
        tm = rng.random(n).astype(np.float32)
        xi = rng.random((n, max(flat.desc.n_media, 1))).astype(np.float32) if flat.desc.n_media else None
        dev = sc.hit_batch(o, d, tm, medium_xi=xi)
        orc = ob.hit_batch(flat, o, d, tm, medium_xi=xi)
        print("   hits:", cmp_hits(dev, orc), flush=True)
        cam = build_camera(cfg, 1.0)
        W = H = 48
        p = abi.make_params(W, H, 4, 50, cfg["background"], seed=11)
        ns = 4000
        px = rng.integers(0, W * H, ns).astype(np.int32); smp = rng.integers(0, 64, ns).astype(np.int32)
        sd = sc.sample_batch(cam, p, px, smp); so = ob.sample_batch(flat, cam, p, px, smp)
        err = np.abs(sd - so).max(axis=1); scale = np.maximum(np.abs(so).max(axis=1), 1e-3)
        bad = err / scale > 1e-3
        print(f"   samples: n={ns} mismatching={int(bad.sum())} max_rel={float((err/scale).max()):.3g} mean_dev={sd.mean():.5f} mean_orc={so.mean():.5f}", flush=True)
        t0 = time.time(); rgb, argb, st = sc.render(cam, p); t_r = time.time() - t0
        orgb, oargb, orays = ob.render(flat, cam, p, nthreads=8)
        derr = np.abs(rgb - orgb).max(axis=2); npx_bad = int((derr > 1e-3 * np.maximum(orgb.max(axis=2), 1e-2)).sum())
        print(f"   render: {t_r*1e3:.1f} ms kernel {st['kernel_ms']:.2f} ms rays dev {st['rays']} orc {orays} bad px {npx_bad}/{W*H} argb mismatch {int((argb != oargb).sum())}", flush=True)
        sc.close()
    ctx.close()

if __name__ == "__main__":
    main()
