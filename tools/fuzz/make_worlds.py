"""Random worlds of the fuzz suite's generator (tests/test_gpu_fuzz.py: every hittable, wrapper, medium, material and
texture kind, nested) written as scene files for tools/fuzz/scene_fuzz: structurally varied VALID input for the
flattener under the sanitizers, where mutated copies of the reference's six worlds mostly vary numbers.
Usage: python tools/fuzz/make_worlds.py <directory> <count> [first seed]"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F  # noqa: E402
from raytracing_amd import flatten, scenefile  # noqa: E402

out, count = sys.argv[1], int(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.makedirs(out, exist_ok=True)
profiles = ["any", "any", "rect", "list", "noise", "instanced"]
sizes = [4, 6, 9, 12, 14, 20, 30, 45, 5, 8, 60, 16, 25, 7, 90, 11, 3, 120, 2, 35]
for s in range(first, first + count):
    name = profiles[s % 6]
    n = sizes[s % 20]
    if name == "instanced":
        n = min(n, 6)
    rng = np.random.default_rng(910000 + s)
    world = F._world(rng, n, float(rng.uniform(4, 12)), F.PROFILES[name])
    flat = flatten.flatten_world(world)
    with open(os.path.join(out, f"w{s:04d}_{name}.rtscn"), "wb") as f:
        f.write(scenefile.dumps(flat))
print("wrote", count, "worlds to", out)
