"""Three ways of filling the idle lanes of the tree walk, priced on the CPU before building any of them: the wave model
of tools/walk_sim.py (final-scene's whole-world tree as the library builds it, real path rays, waves of 53 live rays on
64 lanes, the while-while walk: every lane steps until it holds a leaf or is done, then the wave tests its leaves
together). Cost unit: one step of the walk for the whole wave; a leaf round of one primitive per lane costs 1.6 (the
in-kernel stamps' ratio), each further primitive of a round 1.0.

1. stealing: a lane whose walk has ended takes the OLDEST entry (the largest subtree) off the traversal stack of a lane
   that still walks, together with that lane's ray and closest-so-far, and walks it; results of one ray are merged by
   the reference's tie order at the end. Victims from the whole wave, from the lane's row of 16, its quad or its
   neighbour (cheaper exchanges: LDS, DPP); a steal event is priced at 0.5 to 1.5 steps for the whole wave. The
   closest-so-far of a ray is shared between its lanes at every leaf round.
2. parked leaves: a lane that reaches a leaf parks it (up to Q - 1 of them) and walks on instead of waiting for the
   wave's leaf round; the round then tests up to Q leaves per lane.

No GPU needed. Usage: python tools/steal_sim.py > profiles/<tag>_walk_fill_sim.txt"""
import contextlib
import io
import os
import sys
import time

import numpy as np

sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
with contextlib.redirect_stdout(io.StringIO()):
    import walk_sim as ws
nodes, root, items, pool = ws.nodes1, ws.root1, ws.full, ws.pool2
slab, hit_prim = ws.slab, ws.hit_prim
rng = np.random.default_rng(7)
DONE = -2
LEAF_ROUND, LEAF_MORE = 1.6, 1.0


def inv_of(d):
    with np.errstate(divide='ignore'):
        return np.clip(1.0 / d, -1e30, 1e30)


def walk_step(k, ray, node, stack, tmax, inv):
    o, d = pool[ray[k]]
    _, _, a, b = nodes[node[k]]
    hl, tl = slab(nodes[a][0], nodes[a][1], o, inv[k], 1e-3, tmax[k])
    hr, tr = slab(nodes[b][0], nodes[b][1], o, inv[k], 1e-3, tmax[k])
    if hl and hr:
        if tr < tl:
            stack[k].append(a); node[k] = b
        else:
            stack[k].append(b); node[k] = a
    elif hl:
        node[k] = a
    elif hr:
        node[k] = b
    elif stack[k]:
        node[k] = stack[k].pop()
    else:
        node[k] = DONE


def sim_wave(rays, G=1, steal_cost=0.0, min_idle=1, Q=1):
    """G: lanes a thief may take from (1: no stealing, 2: its neighbour, 4: its quad, 16: its row, 64: the wave)."""
    lanes = rng.permutation(64)[:len(rays)]
    ray = [None] * 64; node = [DONE] * 64; stack = [[] for _ in range(64)]; tmax = [0.0] * 64; owner = list(range(64))
    inv = [None] * 64; pend = [[] for _ in range(64)]
    shared = {}
    for r, k in zip(rays, lanes):
        ray[k] = r; node[k] = root; tmax[k] = 1e30; shared[k] = 1e30
        inv[k] = inv_of(pool[r][1])
        ok, _ = slab(nodes[root][0], nodes[root][1], pool[r][0], inv[k], 1e-3, 1e30)
        if not ok:
            node[k] = DONE
    cost = 0.0; steps = 0; rounds = 0; steals = 0; lane_steps = 0

    def internal(k): return node[k] >= 0 and nodes[node[k]][2] != -1
    def leaf(k): return node[k] >= 0 and nodes[node[k]][2] == -1
    def pop(k): node[k] = stack[k].pop() if stack[k] else DONE

    def do_steal():
        nonlocal cost, steals
        if G == 1 or sum(1 for k in range(64) if node[k] == DONE and not pend[k]) < min_idle:
            return False
        took = False
        for g0 in range(0, 64, G):
            grp = range(g0, g0 + G)
            idle = [k for k in grp if node[k] == DONE and not pend[k]]
            donors = sorted([k for k in grp if node[k] != DONE and stack[k]], key=lambda k: -len(stack[k]))
            for th, dn in zip(idle, donors):
                node[th] = stack[dn].pop(0); ray[th] = ray[dn]; inv[th] = inv[dn]; owner[th] = owner[dn]
                tmax[th] = tmax[dn]; stack[th] = []; took = True
        if took:
            cost += steal_cost; steals += 1
        return took

    while True:
        while True:
            if Q > 1:
                for k in range(64):
                    while leaf(k) and len(pend[k]) < Q - 1:
                        pend[k].append(node[k]); pop(k)
            act = [k for k in range(64) if internal(k)]
            if not act:
                break
            for k in act:
                walk_step(k, ray, node, stack, tmax, inv)
            lane_steps += len(act); cost += 1; steps += 1
            do_steal()
        todo = [pend[k] + ([node[k]] if leaf(k) else []) for k in range(64)]
        m = max(len(t) for t in todo)
        if m == 0:
            if do_steal():
                continue
            break
        rounds += 1; cost += LEAF_ROUND + LEAF_MORE * (m - 1)
        for k in range(64):
            if not todo[k]:
                continue
            o, d = pool[ray[k]]
            for lf in todo[k]:
                for p in items[nodes[lf][3]][2]:
                    h = hit_prim(p, o, d, 1e-3, tmax[k])
                    if h:
                        tmax[k] = h[0]
            shared[owner[k]] = min(shared[owner[k]], tmax[k])
            if leaf(k):
                pop(k)
            pend[k] = []
        for k in range(64):
            if node[k] != DONE:
                tmax[k] = min(tmax[k], shared[owner[k]])
        do_steal()
    return cost, steps, rounds, steals, lane_steps


if __name__ == "__main__":
    waves = rng.permutation(len(pool))[: (len(pool) // 53) * 53].reshape(-1, 53)[:50]
    t0 = time.time()
    print(f"final-scene, {len(nodes)} nodes, {len(waves)} waves of 53 rays; ideal = lane-steps / 64")
    rows = [("the walk as built", {}),
            ("steal from the neighbour, event 0.5", dict(G=2, steal_cost=0.5)),
            ("steal within the quad, event 0.5", dict(G=4, steal_cost=0.5)),
            ("steal within the row of 16, event 0.7", dict(G=16, steal_cost=0.7)),
            ("steal from any lane, event 1.0", dict(G=64, steal_cost=1.0)),
            ("steal from any lane, event 1.5", dict(G=64, steal_cost=1.5)),
            ("steal from any lane, 16 idle, event 1.5", dict(G=64, steal_cost=1.5, min_idle=16)),
            ("one parked leaf (Q = 2)", dict(Q=2)),
            ("two parked leaves (Q = 3)", dict(Q=3)),
            ("three parked leaves (Q = 4)", dict(Q=4))]
    for label, kw in rows:
        r = np.array([sim_wave(list(w), **kw) for w in waves]).mean(0)
        print(f"{label:42s} cost {r[0]:6.2f}  wave-steps {r[1]:6.2f}  leaf rounds {r[2]:5.2f}  steal events {r[3]:5.2f}  "
              f"lane-steps {r[4]:6.1f} (ideal {r[4] / 64:5.2f})  [{time.time() - t0:.0f} s]", flush=True)
