"""Ad-hoc GPU check: hit records for the rays real paths produce (bounce rays that start on
surfaces), device vs oracle; prints the worst offenders."""
import sys
import numpy as np
from raytracing_amd import abi, lib, scenes
from raytracing_amd.core import build_camera, setup
from oracle import binding as ob

def main():
    names = sys.argv[1:] or ["cornell-box", "final-scene", "random-scene", "cornell-smoke"]
    ctx = lib.Context(0)
    rng = np.random.default_rng(5)
    for name in names:
        cfg, flat = setup(name, 3)
        sc = ctx.scene(flat)
        cam = build_camera(cfg, 1.0)
        W = H = 64
        seed = 11
        p = abi.make_params(W, H, 4, 50, cfg["background"], seed=seed)
        rays, xis = [], []
        nm = flat.desc.n_media
        for _ in range(1500):
            px = int(rng.integers(0, W * H)); smp = int(rng.integers(0, 1000))
            r = ob.path_rays(flat, cam, p, px, smp)
            rays.append(r)
            if nm:
                for b in range(len(r)):
                    row = []
                    for m in range(nm):
                        o = ob.philox([m >> 2, (b << 4) | 4, seed & 0xffffffff, seed >> 32], [px, smp])
                        row.append((o[m & 3] >> 8) / 16777216.0)
                    xis.append(row)
        R = np.concatenate(rays)
        o32 = R[:, 0:3].astype(np.float32); d32 = R[:, 3:6].astype(np.float32); t32 = R[:, 6].astype(np.float32)
        xi = np.array(xis, dtype=np.float32) if nm else None
        dev = sc.hit_batch(o32, d32, t32, medium_xi=xi)
        orc = ob.hit_batch(flat, o32, d32, t32, medium_xi=xi)
        hd = dev["hit"].astype(bool); ho = orc[:, 11] > 0.5
        both = hd & ho
        scale_p = np.maximum(np.linalg.norm(orc[:, 1:4], axis=1), 1.0)
        e_t = np.abs(dev["t"] - orc[:, 0]) / np.maximum(np.abs(orc[:, 0]), 1e-3)
        e_p = np.linalg.norm(dev["p"] - orc[:, 1:4], axis=1) / scale_p
        e_n = np.linalg.norm(dev["normal"] - orc[:, 4:7], axis=1) / np.maximum(np.linalg.norm(orc[:, 4:7], axis=1), 1.0)
        e_uv = np.maximum(np.abs(dev["u"] - orc[:, 7]), np.abs(dev["v"] - orc[:, 8]))
        e_uv = np.where(np.isnan(e_uv) & np.isnan(orc[:, 7]) , 0, e_uv)
        bad = (hd != ho) | (both & ((e_t > 1e-5) | (e_p > 1e-5) | (e_n > 1e-5) | (e_uv > 1e-5) |
                                     (dev["front_face"] != orc[:, 9]) | (dev["mat_id"] != orc[:, 10])))
        print(f"== {name}: rays {len(R)} hits {int(both.sum())} bad {int(bad.sum())} "
              f"max t {np.nanmax(e_t[both]):.3g} p {np.nanmax(e_p[both]):.3g} n {np.nanmax(e_n[both]):.3g} uv {np.nanmax(e_uv[both]):.3g}", flush=True)
        for i in np.nonzero(bad)[0][:12]:
            print("  ray o", R[i, 0:3], "d", R[i, 3:6], "tm", R[i, 6])
            print("   dev", dev[i])
            print("   orc", orc[i])
        sc.close()
    ctx.close()

if __name__ == "__main__":
    main()
