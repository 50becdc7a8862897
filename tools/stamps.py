"""Diagnostic: wave-cycle shares of the path loop's phases, from the -DRT_STAMPS build
(make -C raytracing_amd/csrc stamps; RTMI_LIB=raytracing_amd/librtmi_stamps.so)."""
import sys
from raytracing_amd import abi, lib
from raytracing_amd.core import build_camera, setup

scene = sys.argv[1] if len(sys.argv) > 1 else "cornell-box"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg, flat = setup(scene, 1)
cam = build_camera(cfg, 1.0)
ctx = lib.Context(0)
sc = ctx.scene(flat)
p = abi.make_params(W, W, spp, 50, cfg["background"], seed=1)
sc.render(cam, p)
_, _, st = sc.render(cam, p)
ev = list(st["events"].values())
q, c, h, s, it, grp = ev[0], ev[1], ev[2], ev[3], ev[4], ev[5]
tot = q + c + h + s
print(f"{scene} {W}x{W} @{spp}: kernel {st['kernel_ms']:.2f} ms, rays/sample {st['rays'] / st['samples']:.2f}")
print(f"wave-iterations {it}, rays per wave-iteration {st['rays'] / it:.1f} (64 = every lane busy)")
for name, v in (("queue/flush", q), ("camera regen", c), ("closest_hit", h), ("scatter+rng", s)):
    print(f"  {name:14s} {v / it:9.1f} cycles per wave-iteration  {100 * v / tot:5.1f} %")
print(f"  total          {tot / it:9.1f}")
print(f"  of closest_hit: group loops {grp / it:9.1f}, record + unwind {(h - grp) / it:9.1f}")
if ev[11]:
    print(f"  of group loops: BVH walk {ev[9] / it:9.1f}, BVH leaf tests {ev[10] / it:9.1f} (longest lane of each wave); "
          f"leaf rounds per wave-iteration {ev[11] / it:.2f}")
if any(ev[12:20]):
    print("  per surface group (wave cycles per wave-iteration): " + "  ".join(f"g{q} {ev[12 + q] / it:.1f}" for q in range(7) if ev[12 + q])
          + f"  | media {ev[19] / it:.1f}")
mx, mn_inv, sq = ev[6], ev[7], ev[8]
if mx:
    import math
    n_waves = round(it * it / max(sq, 1)) if sq else 0  # exact only when all waves are equal; shown as a hint
    mn = (1 << 64) - 1 - mn_inv
    print(f"  iterations per wave: max {mx}, min {mn}, mean-square root {math.sqrt(sq / max(1, 7168)):.0f} (7168 waves assumed)")
