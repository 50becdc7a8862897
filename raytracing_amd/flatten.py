"""The new host layer: walks a world built by the reference's constructors and writes the flat
scene description of include/rtmi.h (struct rt_scene_desc).

The walk keeps the reference's tree: the world vector and Box sides become LIST nodes in scan
order (base.clj:92-101, box.clj:10-11), BvhNode keeps its left/right/box (base.clj:116-124),
RotateY / Translate / ConstantMedium keep their single child (rotate.clj:9, translate.clj:8,
constant_medium.clj:9). Objects, materials, textures, Perlin tables and images are shared by
identity exactly as the reference shares them (e.g. the `white` Lambertian of scenes.clj:73).
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List

import numpy as np

from . import abi
from . import hittables as H
from . import materials as M
from . import texture as T


class FlatScene:
    """Numpy arrays of one world plus the ctypes struct that points at them."""

    def __init__(self):
        self.prim_type: List[int] = []
        self.prim_d: List[List[float]] = []
        self.prim_mat: List[int] = []
        self.node_kind: List[int] = []
        self.node_i: List[List[int]] = []
        self.node_d: List[List[float]] = []
        self.node_children: List[int] = []
        self.mat_type: List[int] = []
        self.mat_d: List[List[float]] = []
        self.mat_tex: List[int] = []
        self.tex_type: List[int] = []
        self.tex_d: List[List[float]] = []
        self.tex_i: List[List[int]] = []
        self.perlin_ranvec: List[np.ndarray] = []
        self.perlin_perm: List[np.ndarray] = []
        self.images: List[np.ndarray] = []
        self.root = -1
        self.n_media = 0
        self.desc: abi.rt_scene_desc = None  # filled by finish()
        self._keep: list = []

    # -- array finalisation ---------------------------------------------------------------
    def finish(self) -> "FlatScene":
        def arr(x, dt, shape):
            a = np.ascontiguousarray(np.asarray(x, dtype=dt).reshape(shape))
            self._keep.append(a)
            return a
        n_prims, n_nodes = len(self.prim_type), len(self.node_kind)
        n_mats, n_tex = len(self.mat_type), len(self.tex_type)
        self.a_prim_type = arr(self.prim_type, np.int32, (n_prims,))
        self.a_prim_d = arr(self.prim_d, np.float64, (n_prims, abi.RT_PRIM_STRIDE))
        self.a_prim_mat = arr(self.prim_mat, np.int32, (n_prims,))
        self.a_node_kind = arr(self.node_kind, np.int32, (n_nodes,))
        self.a_node_i = arr(self.node_i, np.int32, (n_nodes, abi.RT_NODE_ISTRIDE))
        self.a_node_d = arr(self.node_d, np.float64, (n_nodes, abi.RT_NODE_DSTRIDE))
        self.a_children = arr(self.node_children, np.int32, (len(self.node_children),))
        self.a_mat_type = arr(self.mat_type, np.int32, (n_mats,))
        self.a_mat_d = arr(self.mat_d, np.float64, (n_mats, abi.RT_MAT_DSTRIDE))
        self.a_mat_tex = arr(self.mat_tex, np.int32, (n_mats,))
        self.a_tex_type = arr(self.tex_type, np.int32, (n_tex,))
        self.a_tex_d = arr(self.tex_d, np.float64, (n_tex, abi.RT_TEX_DSTRIDE))
        self.a_tex_i = arr(self.tex_i, np.int32, (n_tex, abi.RT_TEX_ISTRIDE))
        n_perlin = len(self.perlin_ranvec)
        self.a_ranvec = arr(self.perlin_ranvec if n_perlin else [], np.float64, (n_perlin, 256, 3))
        self.a_perm = arr(self.perlin_perm if n_perlin else [], np.int32, (n_perlin, 3, 256))
        n_img = len(self.images)
        self.a_img_w = arr([im.shape[1] for im in self.images], np.int32, (n_img,))
        self.a_img_h = arr([im.shape[0] for im in self.images], np.int32, (n_img,))
        self._img_ptrs = (C.POINTER(C.c_uint8) * max(n_img, 1))()
        for k, im in enumerate(self.images):
            self._img_ptrs[k] = im.ctypes.data_as(C.POINTER(C.c_uint8))

        def p(a, ct):
            return a.ctypes.data_as(C.POINTER(ct))
        d = abi.rt_scene_desc()
        d.abi_version = abi.RTMI_ABI_VERSION
        d.n_prims, d.prim_type, d.prim_d, d.prim_mat = n_prims, p(self.a_prim_type, C.c_int32), p(self.a_prim_d, C.c_double), p(self.a_prim_mat, C.c_int32)
        d.n_nodes, d.node_kind, d.node_i, d.node_d = n_nodes, p(self.a_node_kind, C.c_int32), p(self.a_node_i, C.c_int32), p(self.a_node_d, C.c_double)
        d.n_children, d.node_children = len(self.node_children), p(self.a_children, C.c_int32)
        d.root, d.n_media = self.root, self.n_media
        d.n_mats, d.mat_type, d.mat_d, d.mat_tex = n_mats, p(self.a_mat_type, C.c_int32), p(self.a_mat_d, C.c_double), p(self.a_mat_tex, C.c_int32)
        d.n_tex, d.tex_type, d.tex_d, d.tex_i = n_tex, p(self.a_tex_type, C.c_int32), p(self.a_tex_d, C.c_double), p(self.a_tex_i, C.c_int32)
        d.n_perlin, d.perlin_ranvec, d.perlin_perm = n_perlin, p(self.a_ranvec, C.c_double), p(self.a_perm, C.c_int32)
        d.n_images, d.img_w, d.img_h = n_img, p(self.a_img_w, C.c_int32), p(self.a_img_h, C.c_int32)
        d.img_rgb8 = C.cast(self._img_ptrs, C.POINTER(C.POINTER(C.c_uint8)))
        self.desc = d
        return self


class _Flattener:
    def __init__(self):
        self.fs = FlatScene()
        self.node_of: Dict[int, int] = {}
        self.mat_of: Dict[int, int] = {}
        self.tex_of: Dict[int, int] = {}
        self.perlin_of: Dict[int, int] = {}
        self.image_of: Dict[int, int] = {}
        self._alive: list = []  # keeps id()-keyed objects alive during the walk

    # -- textures / materials ---------------------------------------------------------------
    def texture(self, t) -> int:
        if id(t) in self.tex_of:
            return self.tex_of[id(t)]
        fs = self.fs
        if isinstance(t, T.CheckerTexture):
            odd, even = self.texture(t.odd), self.texture(t.even)
            rec = (abi.RT_TEX_CHECKER, [0.0] * 4, [odd, even])
        elif isinstance(t, T.SolidColor):
            rec = (abi.RT_TEX_SOLID, list(t.color_value) + [0.0], [0, 0])
        elif isinstance(t, T.NoiseTexture):
            pn = t.perlin_noise
            if id(pn) not in self.perlin_of:
                self.perlin_of[id(pn)] = len(fs.perlin_ranvec)
                self._alive.append(pn)
                fs.perlin_ranvec.append(np.asarray(pn.ranvec, dtype=np.float64).reshape(256, 3))
                fs.perlin_perm.append(np.stack([pn.perm_x, pn.perm_y, pn.perm_z]).astype(np.int32))
            rec = (abi.RT_TEX_NOISE, [0.0] * 4, [self.perlin_of[id(pn)], int(t.scale)])
        elif isinstance(t, T.ImageTexture):
            if t.data is None:
                img = -1
            else:
                if id(t.data) not in self.image_of:
                    self.image_of[id(t.data)] = len(fs.images)
                    fs.images.append(np.ascontiguousarray(t.data, dtype=np.uint8))
                img = self.image_of[id(t.data)]
            rec = (abi.RT_TEX_IMAGE, [0.0] * 4, [img, 0])
        else:
            raise TypeError(f"not a texture: {t!r}")
        idx = len(fs.tex_type)
        fs.tex_type.append(rec[0]); fs.tex_d.append(rec[1]); fs.tex_i.append(rec[2])
        self.tex_of[id(t)] = idx
        self._alive.append(t)
        return idx

    def material(self, m) -> int:
        if id(m) in self.mat_of:
            return self.mat_of[id(m)]
        if isinstance(m, M.Lambertian):
            rec = (abi.RT_MAT_LAMBERTIAN, [0.0] * 4, self.texture(m.albedo))
        elif isinstance(m, M.Metal):
            rec = (abi.RT_MAT_METAL, list(m.albedo) + [m.fuzz], -1)
        elif isinstance(m, M.Dielectric):
            rec = (abi.RT_MAT_DIELECTRIC, [m.ir, 0.0, 0.0, 0.0], -1)
        elif isinstance(m, M.DiffuseLight):
            rec = (abi.RT_MAT_DIFFUSE_LIGHT, [0.0] * 4, self.texture(m.emit))
        elif isinstance(m, M.Isotropic):
            rec = (abi.RT_MAT_ISOTROPIC, [0.0] * 4, self.texture(m.albedo))
        else:
            raise TypeError(f"not a material: {m!r}")
        fs = self.fs
        idx = len(fs.mat_type)
        fs.mat_type.append(rec[0]); fs.mat_d.append(rec[1]); fs.mat_tex.append(rec[2])
        self.mat_of[id(m)] = idx
        self._alive.append(m)
        return idx

    # -- hittables --------------------------------------------------------------------------
    def _new_node(self, obj, kind, ni, nd) -> int:
        fs = self.fs
        idx = len(fs.node_kind)
        fs.node_kind.append(kind)
        fs.node_i.append(list(ni) + [0] * (abi.RT_NODE_ISTRIDE - len(ni)))
        fs.node_d.append(list(nd) + [0.0] * (abi.RT_NODE_DSTRIDE - len(nd)))
        if obj is not None:
            self.node_of[id(obj)] = idx
            self._alive.append(obj)
        return idx

    def _prim(self, obj, ptype, params, mat) -> int:
        fs = self.fs
        pidx = len(fs.prim_type)
        fs.prim_type.append(ptype)
        fs.prim_d.append(list(params) + [0.0] * (abi.RT_PRIM_STRIDE - len(params)))
        fs.prim_mat.append(self.material(mat))
        return self._new_node(obj, abi.RT_NODE_PRIM, [pidx], [])

    def _list(self, obj, children) -> int:
        ids = [self.node(c) for c in children]
        first = len(self.fs.node_children)
        self.fs.node_children.extend(ids)
        return self._new_node(obj, abi.RT_NODE_LIST, [first, len(ids)], [])

    def node(self, o) -> int:
        if id(o) in self.node_of:  # shared by identity (BvhNode span 1: left == right)
            return self.node_of[id(o)]
        if isinstance(o, H.Sphere):
            return self._prim(o, abi.RT_PRIM_SPHERE, list(o.center) + [o.radius], o.mat_ptr)
        if isinstance(o, H.MovingSphere):
            return self._prim(o, abi.RT_PRIM_MOVING_SPHERE,
                              list(o.center0) + list(o.center1) + [o.time0, o.time1, o.radius], o.mat_ptr)
        if isinstance(o, H.XYRect):
            return self._prim(o, abi.RT_PRIM_XY_RECT, [o.x0, o.x1, o.y0, o.y1, o.k], o.mp)
        if isinstance(o, H.XZRect):
            return self._prim(o, abi.RT_PRIM_XZ_RECT, [o.x0, o.x1, o.z0, o.z1, o.k], o.mp)
        if isinstance(o, H.YZRect):
            return self._prim(o, abi.RT_PRIM_YZ_RECT, [o.y0, o.y1, o.z0, o.z1, o.k], o.mp)
        if isinstance(o, H.Box):
            return self._list(o, o.sides)
        if isinstance(o, H.HittableList):
            return self._list(o, o.objects)
        if isinstance(o, H.BvhNode):
            left = self.node(o.left)
            right = self.node(o.right)
            return self._new_node(o, abi.RT_NODE_BVH, [left, right],
                                  list(o.box.minimum) + list(o.box.maximum))
        if isinstance(o, H.Translate):
            return self._new_node(o, abi.RT_NODE_TRANSLATE, [self.node(o.ptr)], list(o.offset))
        if isinstance(o, H.RotateY):
            return self._new_node(o, abi.RT_NODE_ROTATE_Y, [self.node(o.ptr)], [o.sin_theta, o.cos_theta])
        if isinstance(o, H.ConstantMedium):
            child = self.node(o.boundary)
            phase = self.material(o.phase_function)
            m = self.fs.n_media
            self.fs.n_media += 1
            return self._new_node(o, abi.RT_NODE_MEDIUM, [child, phase, m], [o.neg_inv_density])
        raise TypeError(f"not a hittable: {o!r}")


def flatten_world(world) -> FlatScene:
    """world: the list a scene constructor returns (or any single hittable)."""
    f = _Flattener()
    if isinstance(world, (list, tuple)):
        f.fs.root = f._list(None, list(world))
    else:
        f.fs.root = f.node(world)
    return f.fs.finish()
