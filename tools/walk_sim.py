"""What the walk of final-scene's large tree (2 400 ground rects, the light, five loose spheres) would cost with
other leaves or more paths per lane: a CPU model of the traversal the kernel does (surface-area sweep tree, near
child first, closest-so-far culling), over real path rays of the world (camera rays and up to seven diffuse
bounces), grouped at random into waves of 53 rays (the measured fill). Prints walk steps and leaf visits per ray
and per wave (a wave walks as long as its slowest lane) for three leaf choices — one rect per leaf (what is built),
one box (six rects) per leaf, five walls per leaf with the lid as a leaf of its own — and the steps per ray slot
when a lane walks K paths one after the other before the wave moves on. No GPU needed.
Usage: python tools/walk_sim.py > profiles/<tag>_walk_sim.txt"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd.core import setup
from raytracing_amd import abi
cfg, flat = setup("final-scene", 1)
pt = np.array(flat.a_prim_type); pd = np.array(flat.a_prim_d).reshape(len(pt), -1)
XY, XZ, YZ = abi.RT_PRIM_XY_RECT, abi.RT_PRIM_XZ_RECT, abi.RT_PRIM_YZ_RECT
rects = [i for i in range(len(pt)) if pt[i] >= XY]
def rect_box(i):
    a0, a1, b0, b1, k = pd[i, :5]; t = pt[i]
    if t == XY: return [a0, b0, k], [a1, b1, k]
    if t == XZ: return [a0, k, b0], [a1, k, b1]
    return [k, a0, b0], [k, a1, b1]
ground = rects[:2400]; light = rects[2400:]
sph = [i for i in range(len(pt)) if pt[i] == 0 and abs(pd[i, 3]) in (50., 70., 100., 80.)]
def sph_box(i):
    c = pd[i, :3]; r = abs(pd[i, 3]); return list(c - r), list(c + r)
# item = (lo, hi, [prims])
def items_A():
    it = [(rect_box(i) + ([i],)) for i in ground + light]
    it += [(sph_box(i) + ([i],)) for i in sph]
    return it
def items_B(merge_lid=True):
    it = []
    for b in range(400):
        six = ground[6 * b: 6 * b + 6]
        los = np.array([rect_box(i)[0] for i in six]); his = np.array([rect_box(i)[1] for i in six])
        if merge_lid:
            it.append((list(los.min(0)), list(his.max(0)), six))
        else:
            lid = [i for i in six if pt[i] == XZ and pd[i, 4] != 0][0]
            rest = [i for i in six if i != lid]
            it.append((list(los.min(0)), list(his.max(0)), rest))
            it.append((rect_box(lid) + ([lid],)))
    it += [(rect_box(i) + ([i],)) for i in light]
    it += [(sph_box(i) + ([i],)) for i in sph]
    return it
def area(lo, hi):
    e = np.maximum(hi - lo, 0); return 2 * (e[0] * e[1] + e[1] * e[2] + e[0] * e[2])
def build(items):
    lo = np.array([i[0] for i in items], float); hi = np.array([i[1] for i in items], float)
    cen = 0.5 * (lo + hi)
    nodes = []  # (lo, hi, left, right) ; leaf: (lo, hi, -1, item)
    def rec(idx):
        if len(idx) == 1:
            nodes.append((lo[idx[0]], hi[idx[0]], -1, idx[0])); return len(nodes) - 1
        best = None
        for ax in range(3):
            o = idx[np.argsort(cen[idx, ax], kind='stable')]
            L_lo = np.minimum.accumulate(lo[o], 0); L_hi = np.maximum.accumulate(hi[o], 0)
            R_lo = np.minimum.accumulate(lo[o][::-1], 0)[::-1]; R_hi = np.maximum.accumulate(hi[o][::-1], 0)[::-1]
            for s in range(1, len(o)):
                c = area(L_lo[s - 1], L_hi[s - 1]) * s + area(R_lo[s], R_hi[s]) * (len(o) - s)
                if best is None or c < best[0]: best = (c, ax, s, o)
        _, ax, s, o = best
        me = len(nodes); nodes.append(None)
        l = rec(o[:s]); r = rec(o[s:])
        nodes[me] = (np.minimum(nodes[l][0], nodes[r][0]), np.maximum(nodes[l][1], nodes[r][1]), l, r)
        return me
    sys.setrecursionlimit(10000)
    root = rec(np.arange(len(items)))
    return nodes, root
def slab(lo, hi, o, inv, tmin, tmax):
    t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = max(np.minimum(t0, t1).max(), tmin); tf = min(np.maximum(t0, t1).min(), tmax)
    return tn <= tf, tn
def hit_prim(i, o, d, tmin, tmax):
    t = pt[i]
    if t >= XY:
        a0, a1, b0, b1, k = pd[i, :5]
        ca, cb, ck = ((0, 1, 2) if t == XY else (0, 2, 1) if t == XZ else (1, 2, 0))
        if d[ck] == 0: return None
        tt = (k - o[ck]) / d[ck]
        if tt < tmin or tt > tmax: return None
        a = o[ca] + tt * d[ca]; b = o[cb] + tt * d[cb]
        if a < a0 or a > a1 or b < b0 or b > b1: return None
        n = np.zeros(3); n[ck] = -1 if d[ck] > 0 else 1
        return tt, n
    c = pd[i, :3]; r = pd[i, 3]; oc = o - c
    a = d @ d; hb = oc @ d; cc = oc @ oc - r * r; disc = hb * hb - a * cc
    if disc < 0: return None
    sq = math.sqrt(disc)
    for root in ((-hb - sq) / a, (-hb + sq) / a):
        if tmin < root < tmax:
            n = (o + root * d - c) / r
            if n @ d > 0: n = -n
            return root, n
    return None
PRIM_WEIGHT = {}  # what a primitive counts in "prim tests" (the last section counts a ground box's six rects as one test)
def trace(nodes, root, items, o, d):
    with np.errstate(divide='ignore'):
        inv = 1.0 / d
    inv = np.clip(inv, -1e30, 1e30)
    tmax = 1e30; best = None; steps = 0; leaves = 0; leaf_prims = 0
    stack = []; node = root
    ok, _ = slab(nodes[root][0], nodes[root][1], o, inv, 1e-3, tmax)
    if not ok: return None, 1, 0, 0
    while True:
        lo, hi, l, r = nodes[node]
        if l == -1:
            leaves += 1
            for p in items[r][2]:
                leaf_prims += PRIM_WEIGHT.get(p, 1.0)
                h = hit_prim(p, o, d, 1e-3, tmax)
                if h: tmax = h[0]; best = h
            if not stack: break
            node = stack.pop(); continue
        steps += 1
        hl, tl = slab(nodes[l][0], nodes[l][1], o, inv, 1e-3, tmax)
        hr, tr = slab(nodes[r][0], nodes[r][1], o, inv, 1e-3, tmax)
        if hl and hr:
            if tr < tl: stack.append(l); node = r
            else: stack.append(r); node = l
        elif hl: node = l
        elif hr: node = r
        elif stack: node = stack.pop()
        else: break
    return best, steps, leaves, leaf_prims
rng = np.random.default_rng(1)
lf = np.array(cfg["lookfrom"], float); la = np.array(cfg["lookat"], float); vfov = cfg["vfov"]
w = lf - la; w /= np.linalg.norm(w); u = np.cross([0, 1, 0], w); u /= np.linalg.norm(u); v = np.cross(w, u)
hh = math.tan(math.radians(vfov) / 2); W, H = 3840, 2160; asp = W / H
def cam_ray(px, py):
    s = (px + rng.random()) / W; t = (py + rng.random()) / H
    d = -w + (2 * s - 1) * hh * asp * u + (2 * t - 1) * hh * v
    return lf.copy(), d
def rand_unit():
    while True:
        p = rng.uniform(-1, 1, 3)
        if p @ p < 1: return p / np.linalg.norm(p)
t0 = time.time()
trees = {}
for name, it in (("rects", items_A()), ("trays", items_B(True)), ("open+lid", items_B(False))):
    nodes, root = build(it); trees[name] = (nodes, root, it)
    print(name, len(it), "items", len(nodes), "nodes", f"{time.time()-t0:.1f}s", flush=True)

pool = []
while len(pool) < 6000:
    o, d = cam_ray(rng.integers(0, W), rng.integers(0, H))
    for b in range(8):
        pool.append((o, d))
        h, *_ = trace(*trees["rects"][:2], trees["rects"][2], o, d)
        if h is None: break
        o = o + h[0] * d; d = h[1] + rand_unit()
print(len(pool), "rays", f"{time.time()-t0:.1f}s", flush=True)
res = {name: np.array([trace(nodes, root, it, o, d)[1:] for o, d in pool]) for name, (nodes, root, it) in trees.items()}
for name, a in res.items():
    idx = rng.permutation(len(pool))[: (len(pool) // 53) * 53].reshape(-1, 53)
    st = a[:, 0][idx]; lv = a[:, 1][idx]
    print(f"{name:9s} steps mean {st.mean():.2f} wave-max {st.max(1).mean():.2f} p99 {np.percentile(a[:,0],99):.0f}  leaves mean {lv.mean():.2f} wave-max {lv.max(1).mean():.2f}  prim tests/ray {a[:,2].mean():.2f}")
print(f"{time.time()-t0:.1f}s")
a = res["rects"][:, 0]
for K in (1, 2, 3, 4):
    n = (len(a) // (53 * K)) * 53 * K
    tot = []
    for rep in range(20):
        x = a[rng.permutation(len(a))[:n]].reshape(-1, 53, K).sum(2).max(1)
        tot.append(x.mean() / K)
    print("paths per lane", K, "walk steps per wave per ray-slot", np.mean(tot))

# ---- leaves of more than one primitive, priced on the tree the library walks now: one leaf per ground box, the light,
# the loose spheres and the thousand r = 10 spheres in world space (scenes.clj:132,144-146). A node whose two children
# are leaves becomes one leaf of both primitives (then of up to four): fewer steps, dearer leaf rounds.
th = math.radians(15.0)
small = [i for i in range(len(pt)) if pt[i] == 0 and abs(pd[i, 3]) == 10.]
def small_box(i):
    x, y, z = pd[i, :3]
    c = np.array([math.cos(th) * x + math.sin(th) * z - 100., y + 270., -math.sin(th) * x + math.cos(th) * z + 395.])
    return list(c - 10.), list(c + 10.), c
world_c = {i: small_box(i)[2] for i in small}
_hit_prim = hit_prim
def hit_prim(i, o, d, tmin, tmax):  # the small spheres are tested where they stand in the world
    if i in world_c:
        c = world_c[i]; oc = o - c
        a = d @ d; hb = oc @ d; cc = oc @ oc - 100.; disc = hb * hb - a * cc
        if disc < 0: return None
        sq = math.sqrt(disc)
        for root in ((-hb - sq) / a, (-hb + sq) / a):
            if tmin < root < tmax:
                n = (o + root * d - c) / 10.
                if n @ d > 0: n = -n
                return root, n
        return None
    return _hit_prim(i, o, d, tmin, tmax)
def collapse(nodes, items, limit, only_spheres):
    nodes = list(nodes); items = list(items)
    for n, nd in enumerate(nodes):
        lo, hi, l, r = nd
        if l == -1 or nodes[l][2] != -1 or nodes[r][2] != -1: continue
        pl = items[nodes[l][3]][2]; pr = items[nodes[r][3]][2]
        if len(pl) + len(pr) > limit: continue
        if only_spheres and any(pt[p] != 0 for p in pl + pr): continue
        items.append((lo, hi, pl + pr)); nodes[n] = (lo, hi, -1, len(items) - 1)
    return nodes, items
t0 = time.time()
full = items_B(True) + [(small_box(i)[:2] + ([i],)) for i in small]
nodes1, root1 = build(full)
print(f"\nwhole world: {len(full)} leaves, {len(nodes1)} nodes {time.time()-t0:.1f}s", flush=True)
pool2 = []
while len(pool2) < 6000:
    o, d = cam_ray(rng.integers(0, W), rng.integers(0, H))
    for b in range(8):
        pool2.append((o, d))
        h, *_ = trace(nodes1, root1, full, o, d)
        if h is None: break
        o = o + h[0] * d; d = h[1] + rand_unit()
PRIM_WEIGHT.update({i: 1.0 / 6.0 for i in ground})
variants = {"one primitive per leaf": (nodes1, full)}
n2, i2 = collapse(nodes1, full, 2, True); variants["pairs of spheres"] = (n2, i2)
n2a, i2a = collapse(nodes1, full, 2, False); variants["pairs of anything"] = (n2a, i2a)
n4, i4 = collapse(n2a, i2a, 4, False); variants["up to four"] = (n4, i4)
# a leaf round of one primitive costs about 1.6 steps of the walk (in-kernel stamps: 1.2 k cycles a step, 1.9 k a round);
# each further primitive of a round is priced at one more step
idx = rng.permutation(len(pool2))[: (len(pool2) // 53) * 53].reshape(-1, 53)  # the same waves for every variant
for name, (nd, it) in variants.items():
    a = np.array([trace(nd, root1, it, o, d)[1:] for o, d in pool2])
    st = a[:, 0][idx].max(1).mean(); lv = a[:, 1][idx].max(1).mean()
    per_round = a[:, 2].sum() / max(a[:, 1].sum(), 1)
    cost = st + lv * (1.6 + (per_round - 1.0))
    print(f"{name:24s} steps/ray {a[:,0].mean():5.2f} wave-max {st:5.2f}  leaf rounds wave-max {lv:4.2f}  tests per leaf visit {per_round:4.2f}"
          f"  priced {cost:5.1f} step-units per wave-iteration")
print(f"{time.time()-t0:.1f}s")
