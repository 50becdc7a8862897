import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup
name = sys.argv[1] if len(sys.argv) > 1 else "random-scene"
ctx = lib.Context(0)
cfg, flat = setup(name, 1)
W, H = 61, 45
cam = build_camera(cfg, W / H)
sc = ctx.scene(flat)
print(sc.info(), sc.variants(), flush=True)
bad = []
for k in range(7):
    p = abi.make_params(W, H, 1, 50, cfg["background"], seed=5, sample_offset=k)
    rgb, _, st = sc.render(cam, p)
    p2 = abi.make_params(W, H, 1, 50, cfg["background"], seed=5, sample_offset=k, flags=abi.RT_FLAG_LANE_PATHS)
    old, _, st2 = sc.render(cam, p2)
    px = np.arange(W * H, dtype=np.int32)
    ref = sc.sample_batch(cam, p, px, np.full(W * H, k, dtype=np.int32)).reshape(H, W, 3)
    d = np.abs(rgb - ref).max(axis=2)
    ys, xs = np.nonzero(d > 0)
    print("sample", k, "rays", st["rays"], "differing pixels", len(ys), "| old kernel rays", st2["rays"], "differing", int((np.abs(old - ref).max(axis=2) > 0).sum()), flush=True)
    for y, x in list(zip(ys, xs))[:3]:
        print("   px", x, y, "pool", rgb[y, x], "ref", ref[y, x])
        bad.append((int(y * W + x), k))
# follow one bad path with the oracle's ray list and the scatter hook
if bad:
    from oracle import binding as ob
    pxi, k = bad[0]
    p = abi.make_params(W, H, 1, 50, cfg["background"], seed=5)
    r = ob.path_rays(flat, cam, p, pxi, k)
    print("oracle path of", pxi, k, "has", len(r), "rays")
    h = sc.hit_batch(r[:, 0:3], r[:, 3:6], r[:, 6])
    for i in range(len(r)):
        print(i, r[i], "hit", h[i]["hit"], "t", h[i]["t"], "mat", h[i]["mat_id"])

print("---- depth bisect")
for pxi, k in bad[:6]:
    p = abi.make_params(W, H, 1, 50, cfg["background"], seed=5)
    r = ob.path_rays(flat, cam, p, pxi, k)
    first = None
    for depth in range(1, 12):
        pd = abi.make_params(W, H, 1, depth, cfg["background"], seed=5, sample_offset=k)
        a, _, _ = sc.render(cam, pd)
        b = sc.sample_batch(cam, pd, np.array([pxi], np.int32), np.array([k], np.int32))[0]
        if not np.array_equal(a.reshape(-1, 3)[pxi], b):
            first = depth
            break
    h = sc.hit_batch(r[:, 0:3], r[:, 3:6], r[:, 6])
    sca = sc.scatter_batch(5, r[:, 0:3], r[:, 3:6], r[:, 6], pxi, k, np.arange(len(r)))
    print("pixel", pxi, "sample", k, "path length", len(r), "first differing depth", first)
    for i in range(len(r)):
        mt = flat.mat_type[h[i]["mat_id"]] if h[i]["hit"] else -1
        print("   ray", i, "hit", h[i]["hit"], "t", h[i]["t"], "mat", h[i]["mat_id"], "mtype", mt, "front", h[i]["front_face"], "|d|", np.linalg.norm(r[i, 3:6]))
