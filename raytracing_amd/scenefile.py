"""Scene files: a FlatScene (struct rt_scene_desc of include/rtmi.h) as one little-endian blob.

The reference has no scene format: its `world` vector lives in the JVM (core.clj:66-88). A host
that cannot pass nested pointers comfortably writes these bytes instead and hands the library one
buffer or one path (rt_scene_create_from_memory / rt_scene_create_from_file). Layout:

    0   magic "RTMISCN1"
    8   int32 abi_version
    12  int32 n_prims n_nodes n_children root n_media n_mats n_tex n_perlin n_images
    48  int32 reserved[4] = 0
    64  the arrays of rt_scene_desc in struct order, then the images (h*w*3 bytes each); every
        array and image zero-padded to a multiple of 8 bytes

This module is an independent writer and reader of that layout in numpy (the library has its own
in C, rt_scene_write); the tests hold the two against each other byte for byte.
"""
from __future__ import annotations

import struct

import numpy as np

from . import abi
from .flatten import FlatScene

MAGIC = b"RTMISCN1"
HEADER_BYTES = 64


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


def _arrays(fs: FlatScene):
    return [fs.a_prim_type, fs.a_prim_d, fs.a_prim_mat, fs.a_node_kind, fs.a_node_i, fs.a_node_d,
            fs.a_children, fs.a_mat_type, fs.a_mat_d, fs.a_mat_tex, fs.a_tex_type, fs.a_tex_d,
            fs.a_tex_i, fs.a_ranvec, fs.a_perm, fs.a_img_w, fs.a_img_h]


def dumps(fs: FlatScene) -> bytes:
    """FlatScene (finished) -> file bytes."""
    d = fs.desc
    head = MAGIC + struct.pack("<14i", d.abi_version, d.n_prims, d.n_nodes, d.n_children, d.root,
                               d.n_media, d.n_mats, d.n_tex, d.n_perlin, d.n_images, 0, 0, 0, 0)
    assert len(head) == HEADER_BYTES
    parts = [head]
    for a in _arrays(fs):
        parts.append(_pad8(np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<"), copy=False).tobytes()))
    for im in fs.images:
        parts.append(_pad8(np.ascontiguousarray(im, dtype=np.uint8).tobytes()))
    return b"".join(parts)


def save(fs: FlatScene, path: str) -> int:
    data = dumps(fs)
    with open(path, "wb") as f:
        f.write(data)
    return len(data)


def loads(data: bytes) -> FlatScene:
    """File bytes -> finished FlatScene. Raises ValueError on a malformed blob."""
    if len(data) < HEADER_BYTES or data[:8] != MAGIC:
        raise ValueError("not a scene file (bad magic or shorter than its header)")
    (version, n_prims, n_nodes, n_children, root, n_media, n_mats, n_tex, n_perlin,
     n_images) = struct.unpack_from("<10i", data, 8)
    if version != abi.RTMI_ABI_VERSION:
        raise ValueError(f"scene file has abi_version {version}, this package speaks {abi.RTMI_ABI_VERSION}")
    if min(n_prims, n_nodes, n_children, n_media, n_mats, n_tex, n_perlin, n_images) < 0:
        raise ValueError("scene file: negative count")
    pos = HEADER_BYTES

    def take(dtype, shape):
        nonlocal pos
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if pos + n > len(data):
            raise ValueError("scene file truncated")
        a = np.frombuffer(data, dtype=np.dtype(dtype).newbyteorder("<"), count=int(np.prod(shape)), offset=pos)
        pos += n + (-n % 8)
        return a.reshape(shape).astype(dtype)

    fs = FlatScene()
    fs.prim_type = take(np.int32, (n_prims,)).tolist()
    fs.prim_d = take(np.float64, (n_prims, abi.RT_PRIM_STRIDE)).tolist()
    fs.prim_mat = take(np.int32, (n_prims,)).tolist()
    fs.node_kind = take(np.int32, (n_nodes,)).tolist()
    fs.node_i = take(np.int32, (n_nodes, abi.RT_NODE_ISTRIDE)).tolist()
    fs.node_d = take(np.float64, (n_nodes, abi.RT_NODE_DSTRIDE)).tolist()
    fs.node_children = take(np.int32, (n_children,)).tolist()
    fs.mat_type = take(np.int32, (n_mats,)).tolist()
    fs.mat_d = take(np.float64, (n_mats, abi.RT_MAT_DSTRIDE)).tolist()
    fs.mat_tex = take(np.int32, (n_mats,)).tolist()
    fs.tex_type = take(np.int32, (n_tex,)).tolist()
    fs.tex_d = take(np.float64, (n_tex, abi.RT_TEX_DSTRIDE)).tolist()
    fs.tex_i = take(np.int32, (n_tex, abi.RT_TEX_ISTRIDE)).tolist()
    ranvec = take(np.float64, (n_perlin, 256, 3))
    perm = take(np.int32, (n_perlin, 3, 256))
    fs.perlin_ranvec = [ranvec[k] for k in range(n_perlin)]
    fs.perlin_perm = [perm[k] for k in range(n_perlin)]
    img_w = take(np.int32, (n_images,))
    img_h = take(np.int32, (n_images,))
    for w, h in zip(img_w.tolist(), img_h.tolist()):
        if w <= 0 or h <= 0:
            raise ValueError("scene file: image with a non-positive size")
        fs.images.append(np.ascontiguousarray(take(np.uint8, (h, w, 3))))
    fs.root, fs.n_media = root, n_media
    return fs.finish()


def load(path: str) -> FlatScene:
    with open(path, "rb") as f:
        return loads(f.read())
