"""Host logic on the CPU: scene mirrors, flattener, ABI layout, library exports."""
import ctypes as C
import os
import random
import re
import subprocess
import tempfile

import numpy as np
import pytest

from raytracing_amd import abi, flatten, lib, scenes
from raytracing_amd.hittables import (BvhNode, Sphere, box, bvhnode, hittable_list, sphere)
from raytracing_amd.materials import lambertian

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GREY = lambertian([0.5, 0.5, 0.5])


def test_scene_inventory():
    # scenes.clj: world sizes as the reference builds them
    rng = random.Random(3)
    assert len(scenes.random_scene(rng)) == 22 * 22 + 5
    assert len(scenes.cornell_box()) == 8 and len(scenes.cornell_smoke()) == 8
    assert len(scenes.final_scene(random.Random(3))) == 11
    assert len(scenes.simple_light(random.Random(3))) == 3 and len(scenes.earth()) == 1
    fs = flatten.flatten_world(scenes.cornell_box())
    assert fs.desc.n_prims == 18 and fs.desc.n_mats == 4 and fs.desc.n_media == 0
    assert list(fs.a_node_kind).count(abi.RT_NODE_TRANSLATE) == 2
    assert list(fs.a_node_kind).count(abi.RT_NODE_ROTATE_Y) == 2
    fs = flatten.flatten_world(scenes.cornell_smoke())
    assert fs.desc.n_media == 2 and fs.desc.n_mats == 6  # 4 + two isotropic phase functions
    cfg = scenes.scene_config("cornell-smoke")
    assert cfg["background"] == (0.70, 0.80, 1.0) and cfg["vfov"] == 40.0   # :or defaults, core.clj:61-65,80-83
    with pytest.raises(KeyError):
        scenes.scene_config("no-such-scene")


def test_random_scene_order_and_ground_tie():
    # z-fix is a stable sort: the checker ground (listed first) stays ahead of the grey ground
    w = scenes.random_scene(random.Random(5))
    grounds = [o for o in w if o.radius == 1000]
    assert len(grounds) == 2
    from raytracing_amd.texture import CheckerTexture, SolidColor
    assert isinstance(grounds[0].mat_ptr.albedo, CheckerTexture) and isinstance(grounds[1].mat_ptr.albedo, SolidColor)
    d = [scenes.v_dist((13.0, 2.0, 3.0), o.center) for o in w]
    assert d == sorted(d)


def test_final_scene_box_tops_are_truncated():
    w = scenes.final_scene(random.Random(3))
    root = w[0]
    leaves = []

    def walk(n):
        if isinstance(n, BvhNode):
            walk(n.left); walk(n.right)
        else:
            leaves.append(n)
    walk(root)
    boxes = {id(b): b for b in leaves}
    assert len(boxes) == 400                 # x is pre-sorted and all min-y are equal: nothing dropped
    b = next(iter(boxes.values()))
    top = b.sides[2]
    assert top.k == float(int(b.box_max[1])) and b.box_max[1] != top.k   # ^long k, box.clj:20
    assert b.sides[4].y1 == b.box_max[1]                                  # sides keep the fraction


def test_bvhnode_mirror_reproduces_the_index_quirk():
    # base.clj:142-161: each level re-sorts the whole vector along its own axis, then uses index
    # sub-ranges; with mixed axes objects are lost or repeated.
    class AxisRng:
        def __init__(self, axes): self.axes = list(axes)
        def random(self): return 0.25 if self.axes.pop(0) == 0 else 0.75
    objs = [sphere([x, 5 - x, 0], 0.1, GREY) for x in range(6)]

    def leaves(n):
        return leaves(n.left) + leaves(n.right) if isinstance(n, BvhNode) else [n]
    # draws in call order: root x | left y, its two children | right x, its two children.
    # root: sorted by x, mid = ceil(0 + 6/2) = 3. The left call re-sorts all six by y and takes
    # indices 0..2 of THAT order (x = 5, 4, 3); the right call re-sorts by x and takes indices
    # 3..5 (x = 3, 4, 5): objects x = 0, 1, 2 are lost and x = 3, 4, 5 appear twice.
    tree = bvhnode(hittable_list(objs), 0, 1, AxisRng([0, 1, 0, 0, 0, 0, 0]))
    got = [o.center[0] for o in leaves(tree)]
    assert got == [4.0, 5.0, 3.0, 3.0, 3.0, 4.0, 5.0, 5.0]
    fs = flatten.flatten_world([tree])
    assert fs.desc.n_prims == 3              # reachable leaves, shared by identity
    # same axis everywhere: a plain median split, nothing lost
    tree = bvhnode(hittable_list(objs), 0, 1, AxisRng([0] * 7))
    assert sorted(set(o.center[0] for o in leaves(tree))) == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
    # three objects: mid = ceil(0 + 3/2) = 2 -> left span 2, right span 1 (both children the same object)
    tree = bvhnode(hittable_list(objs[:3]), 0, 1, AxisRng([0, 0, 0]))
    assert tree.right.left is tree.right.right
    assert flatten.flatten_world([tree]).desc.n_prims == 3


def test_flattener_shares_by_identity():
    fs = flatten.flatten_world(scenes.two_spheres())
    assert fs.desc.n_mats == 2 and fs.desc.n_tex == 3   # two Lambertians over one checker (scenes.clj:29-32)
    b = box([0, 0, 0], [1, 2.5, 3], GREY)
    fs = flatten.flatten_world([b, b])
    assert fs.desc.n_prims == 6
    kinds = list(fs.a_node_kind)
    assert kinds.count(abi.RT_NODE_LIST) == 2 and fs.a_children.tolist()[-2:] == [6, 6]
    # box.clj:18-23 side order and the truncated k of the top face
    assert list(fs.a_prim_type) == [abi.RT_PRIM_XY_RECT] * 2 + [abi.RT_PRIM_XZ_RECT] * 2 + [abi.RT_PRIM_YZ_RECT] * 2
    assert fs.a_prim_d[2, 4] == 2.0 and fs.a_prim_d[4, 1] == 2.5


def _c_layout():
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "rtmi.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(rt_scene_desc), sizeof(rt_camera), sizeof(rt_params),
         sizeof(rt_stats), sizeof(rt_hit), sizeof(rt_scene_info));
  printf("%zu %zu %zu %zu\n", offsetof(rt_scene_desc, img_rgb8), offsetof(rt_params, seed),
         offsetof(rt_stats, kernel_ms), offsetof(rt_hit, mat_id));
  printf("%d\n", RT_EV_COUNT);
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "l.c")
        with open(c, "w") as f:
            f.write(src)
        exe = os.path.join(td, "l")
        subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        return subprocess.check_output([exe]).decode().split()


def test_ctypes_layout_matches_the_c_header():
    out = [int(x) for x in _c_layout()]
    assert out[:6] == [C.sizeof(abi.rt_scene_desc), C.sizeof(abi.rt_camera), C.sizeof(abi.rt_params),
                       C.sizeof(abi.rt_stats), C.sizeof(abi.rt_hit), C.sizeof(abi.rt_scene_info)]
    assert out[6:10] == [abi.rt_scene_desc.img_rgb8.offset, abi.rt_params.seed.offset,
                         abi.rt_stats.kernel_ms.offset, abi.rt_hit.mat_id.offset]
    assert out[10] == abi.RT_EV_COUNT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    declared = sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(lib.EXPORTS)
    L = lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.rt_abi_version() == abi.RTMI_ABI_VERSION


def test_no_cpu_fallback():
    """Without a HIP device the product path must fail loudly, not fall back to anything."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the failure path cannot be exercised")
    with pytest.raises(lib.RtError) as ei:
        lib.Context(0)
    assert ei.value.code == abi.RT_ERR_HIP and "no CPU path" in str(ei.value)


def test_product_package_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "raytracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.lower().replace("no cpu", ""), f"{fn} mentions the oracle"


def test_buffered_image_placement_and_ppm(tmp_path):
    from raytracing_amd.core import to_buffered_image, write_ppm
    argb = np.zeros((4, 3), dtype=np.uint32)
    argb[0, 0] = 0xFF112233   # y = 0 is the bottom row
    img = to_buffered_image(argb)
    assert img.shape == (4 + 75, 3 + 20) and img[4 + 30, 0] == 0x112233   # core.clj:92-94,123-126
    path = str(tmp_path / "o.ppm")
    write_ppm(path, argb)
    data = open(path, "rb").read()
    assert data.startswith(b"P6\n3 4\n255\n") and data[-9:-6] == bytes([0x11, 0x22, 0x33])
