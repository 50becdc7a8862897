"""Parity report for every reference world: hit records of probe rays (device FP32 against the FP64
CPU restatement) and full renders with INDEPENDENT random streams on the two sides (per-pixel z
scores against the Monte Carlo error, RMSE, PSNR). Run on the GPU box:
    PYTHONPATH=. python tools/parity_report.py > profiles/parity.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob  # noqa: E402
from raytracing_amd import abi, lib  # noqa: E402
from util import ALL_SCENES, compare_hits, medium_uniform, probe_rays, scene_bundle  # noqa: E402

W = H = 96
BATCHES, SPP = 6, 64
ctx = lib.Context(0)
print(f"{'world':20s} {'rays':>6s} {'hit!=':>5s} {'>1e-5':>5s} {'dt/t':>9s} {'dp':>9s} {'dn':>9s} {'duv':>9s} |"
      f" {'|z|>3':>6s} {'RMSE':>8s} {'PSNR dB':>8s} {'mean dev':>9s} {'mean cpu':>9s}")
for name in ALL_SCENES:
    cfg, flat, cam = scene_bundle(name)
    sc = ctx.scene(flat)
    o32, d32, tm = probe_rays(cfg, 20000, 11)
    nm = flat.desc.n_media
    xi = np.random.default_rng(12).random((len(o32), max(nm, 1))).astype(np.float32) if nm else None
    dev = sc.hit_batch(o32, d32, tm, medium_xi=xi)
    orc = ob.hit_batch(flat, o32, d32, tm, medium_xi=xi)
    e = compare_hits(dev, orc, d32)
    b = e["both"]
    t_floor = np.linalg.norm(o32.astype(np.float64), axis=1) / np.maximum(np.linalg.norm(d32.astype(np.float64), axis=1), 1e-30)
    t_err = np.abs(dev["t"] - orc[:, 0]) / np.maximum(np.abs(orc[:, 0]), t_floor)
    d_img = np.stack([sc.render(cam, abi.make_params(W, H, SPP, 50, cfg["background"], seed=1000 + k))[0] / SPP for k in range(BATCHES)])
    c_img = np.stack([ob.render(flat, cam, abi.make_params(W, H, SPP, 50, cfg["background"], seed=2000 + k), nthreads=16, want_argb=False)[0] / SPP
                      for k in range(BATCHES)])
    md, mc = d_img.mean(axis=0), c_img.mean(axis=0)
    se = np.sqrt(d_img.var(axis=0, ddof=1) / BATCHES + c_img.var(axis=0, ddof=1) / BATCHES)
    z = np.abs(md - mc) / np.maximum(se, 1e-6)
    rmse = float(np.sqrt(((md - mc) ** 2).mean()))
    psnr = 10 * np.log10(max(float(mc.max()), 1e-6) ** 2 / max(rmse ** 2, 1e-20))
    q = lambda a: float(np.nanquantile(a[b], 0.9999)) if b.any() else 0.0
    over = int((b & ((t_err > 1e-5) | (e["p"] > 1e-5) | (e["n"] > 1e-5) | (e["uv"] > 1e-5) | e["front"] | e["mat"])).sum())
    print(f"{name:20s} {len(o32):6d} {e['hit_mismatch']:5d} {over:5d} {q(t_err):9.2e} {q(e['p']):9.2e} "
          f"{q(e['n']):9.2e} {q(e['uv']):9.2e} | {float((z > 3).mean()):6.4f} {rmse:8.4f} {psnr:8.1f} "
          f"{md.mean():9.5f} {mc.mean():9.5f}", flush=True)
    sc.close()
print(f"renders: {W}x{H}, {BATCHES} batches x {SPP} spp per side, different seeds on the two sides; z = |mean difference| / standard error "
      "(t-distributed, 10 dof: P(|z|>3) = 1.3 %). hit records: 20000 probe rays per world; error columns are the 99.99th percentile (relative as in tests/test_gpu_parity.py), '>1e-5' counts the rays beyond the bar in any field (knife-edge rays; final-scene: u/v on coincident faces of adjacent ground boxes, which the reference resolves through the shape of its own BVH).")

# The headline world at a higher sample count: 256 x 256, 8 batches x 256 spp on each side (2 048 spp
# per side), independent streams; the z scores are then t-distributed with 14 degrees of freedom
# (P(|z| > 3) = 1.0 %).
cfg, flat, cam = scene_bundle("cornell-box")
sc = ctx.scene(flat)
W2 = H2 = 256
B2, S2 = 8, 256
d_img = np.stack([sc.render(cam, abi.make_params(W2, H2, S2, 50, cfg["background"], seed=3000 + k))[0] / S2 for k in range(B2)])
c_img = np.stack([ob.render(flat, cam, abi.make_params(W2, H2, S2, 50, cfg["background"], seed=4000 + k), nthreads=16, want_argb=False)[0] / S2
                  for k in range(B2)])
md, mc = d_img.mean(axis=0), c_img.mean(axis=0)
se = np.sqrt(d_img.var(axis=0, ddof=1) / B2 + c_img.var(axis=0, ddof=1) / B2)
z = np.abs(md - mc) / np.maximum(se, 1e-6)
rmse = float(np.sqrt(((md - mc) ** 2).mean()))
noise = float(np.sqrt((se ** 2).mean()))
psnr = 10 * np.log10(max(float(mc.max()), 1e-6) ** 2 / max(rmse ** 2, 1e-20))
print(f"cornell-box {W2}x{H2}, {B2} x {S2} spp per side: |z|>3 {float((z > 3).mean()):.4f} (expected 0.010), |z|>4 {float((z > 4).mean()):.5f} "
      f"(expected 0.0013), RMSE {rmse:.4f} against a Monte Carlo standard error of {noise:.4f} (root mean square over pixels), "
      f"PSNR {psnr:.1f} dB, image means {md.mean():.5f} (device) {mc.mean():.5f} (CPU restatement)")
sc.close()
