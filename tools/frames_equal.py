"""Usage (GPU box): python tools/frames_equal.py <tag a> <tag b>  (libraries ab/librtmi_<tag>.so from tools/ab_build.sh)
Renders a few worlds with two libraries (RTMI_LIB) in child processes and compares the packed frames byte for byte."""
import hashlib, os, subprocess, sys
code = r'''
import hashlib, sys
import numpy as np
from raytracing_amd import abi, lib, flatten
from raytracing_amd.core import build_camera
from raytracing_amd import scenes
ctx = lib.Context(0)
for name in sys.argv[1:]:
    from util import scene_bundle
    cfg, flat, cam = scene_bundle(name)
    sc = ctx.scene(flat)
    p = abi.make_params(192, 192, 24, 50, cfg["background"], seed=77)
    frame = sc.render(cam, p)[0]
    print(name, hashlib.sha256(np.ascontiguousarray(frame).tobytes()).hexdigest())
'''
res = {}
for tag in sys.argv[1:3]:
    env = dict(os.environ, RTMI_LIB=f"ab/librtmi_{tag}.so", PYTHONPATH=".:tests")
    res[tag] = subprocess.run([sys.executable, "-c", code, "two-perlin-spheres", "simple-light", "final-scene", "two-spheres", "random-scene"], env=env, capture_output=True, text=True)
    print(tag, res[tag].stdout, res[tag].stderr[-400:])
a, b = (res[t].stdout for t in sys.argv[1:3])
print("EQUAL" if a == b and a.strip() else "DIFFERENT")
