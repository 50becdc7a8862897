"""ctypes binding of librtmi.so (include/rtmi.h). There is no CPU path: if the shared library
or a HIP device is missing, loading or `Context()` raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTMI_LIB", os.path.join(_HERE, "librtmi.so"))

EXPORTS = ["rt_init", "rt_shutdown", "rt_last_error", "rt_abi_version", "rt_scene_create",
           "rt_scene_destroy", "rt_scene_get_info", "rt_render", "rt_render_device",
           "rt_tonemap_device", "rt_hit_batch", "rt_sample_batch", "rt_scene_file_size",
           "rt_scene_write", "rt_scene_create_from_memory", "rt_scene_create_from_file",
           "rt_scene_create_with", "rt_scene_bvh_digest", "rt_scene_get_variants", "rt_hit_batch_variant", "rt_scatter_batch",
           "rt_render_device_noalloc", "rt_get_device_info", "rt_multi_init", "rt_multi_shutdown",
           "rt_multi_device_count", "rt_multi_selftest_rccl", "rt_multi_last_error", "rt_multi_scene_create", "rt_multi_scene_destroy",
           "rt_multi_render", "rt_image_decode_jpeg", "rt_image_decode_png", "rt_image_decode_gif", "rt_image_decode_bmp",
           "rt_image_decode"]

_LIB = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtmi error {code}: {msg}")
        self.code = code


def load():
    """dlopen librtmi.so and declare its prototypes. Raises if the library was not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch bundles its own libamdhip64/libhsa-runtime64. If it is
    # going to be used in this process (device buffers, torch.distributed), it has to be loaded
    # first so that librtmi.so binds to the same runtime; a second HSA runtime finds no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    fp, ip, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    L.rt_init.argtypes = [C.c_int32, C.POINTER(vp)]
    L.rt_shutdown.argtypes = [vp]
    L.rt_shutdown.restype = None
    L.rt_last_error.restype = C.c_char_p
    L.rt_abi_version.restype = C.c_int
    L.rt_scene_create.argtypes = [vp, C.POINTER(abi.rt_scene_desc), C.POINTER(vp)]
    L.rt_scene_create_with.argtypes = [vp, C.POINTER(abi.rt_scene_desc), C.c_int32, C.POINTER(vp)]
    L.rt_scene_get_variants.argtypes = [vp, ip, u32p]
    L.rt_scene_bvh_digest.argtypes = [vp, C.POINTER(C.c_uint64), ip, ip]
    L.rt_hit_batch_variant.argtypes = [vp, vp, C.c_int32, C.c_int32, fp, fp, fp, C.c_float, C.c_float, fp,
                                       C.POINTER(abi.rt_hit)]
    L.rt_scatter_batch.argtypes = [vp, vp, C.c_uint64, C.c_int32, fp, fp, fp, ip, ip, ip, C.POINTER(abi.rt_scatter)]
    for f in (L.rt_image_decode_jpeg, L.rt_image_decode_png, L.rt_image_decode_gif, L.rt_image_decode_bmp, L.rt_image_decode):
        f.argtypes = [C.c_char_p, C.c_uint64, ip, ip, vp, C.c_uint64]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_get_device_info.argtypes = [vp, C.POINTER(abi.rt_device_info)]
    L.rt_multi_init.argtypes = [ip, C.c_int32, C.POINTER(vp)]
    L.rt_multi_shutdown.argtypes = [vp]
    L.rt_multi_shutdown.restype = None
    L.rt_multi_device_count.argtypes = [vp]
    L.rt_multi_selftest_rccl.argtypes = [C.c_int32]
    L.rt_multi_last_error.restype = C.c_char_p
    L.rt_multi_scene_create.argtypes = [vp, C.POINTER(abi.rt_scene_desc), C.POINTER(vp)]
    L.rt_multi_scene_destroy.argtypes = [vp]
    L.rt_multi_scene_destroy.restype = None
    L.rt_multi_render.argtypes = [vp, vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), C.c_int32, fp, u32p,
                                  C.POINTER(abi.rt_stats)]
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(abi.rt_scene_info)]
    L.rt_render.argtypes = [vp, vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), fp, u32p,
                            C.POINTER(abi.rt_stats)]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), vp, vp,
                                   C.POINTER(abi.rt_stats), vp]
    L.rt_render_device_noalloc.argtypes = [vp, vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), vp, vp, vp]
    L.rt_tonemap_device.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.rt_hit_batch.argtypes = [vp, vp, C.c_int32, fp, fp, fp, C.c_float, C.c_float, fp, C.POINTER(abi.rt_hit)]
    L.rt_sample_batch.argtypes = [vp, vp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), C.c_int32,
                                  ip, ip, fp]
    L.rt_scene_file_size.argtypes = [C.POINTER(abi.rt_scene_desc)]
    L.rt_scene_file_size.restype = C.c_uint64
    L.rt_scene_write.argtypes = [C.POINTER(abi.rt_scene_desc), vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.rt_scene_create_from_memory.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp)]
    L.rt_scene_create_from_file.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise RtError(rc, load().rt_last_error().decode("utf-8", "replace"))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def scene_file_bytes(flat) -> bytes:
    """The scene file of a FlatScene as the library's own writer produces it (rt_scene_write;
    needs no device)."""
    L = load()
    n = L.rt_scene_file_size(C.byref(flat.desc))
    if n == 0:
        raise RtError(abi.RT_ERR_ARG, L.rt_last_error().decode("utf-8", "replace"))
    buf = (C.c_uint8 * n)()
    written = C.c_uint64(0)
    _check(L.rt_scene_write(C.byref(flat.desc), buf, n, C.byref(written)))
    return bytes(buf[:written.value])


def stats_dict(st: abi.rt_stats) -> dict:
    return dict(samples=int(st.samples), rays=int(st.rays), kernel_ms=float(st.kernel_ms), shader_mhz=float(st.shader_mhz),
                events={n: int(st.events[i]) for i, n in enumerate(abi.RT_EVENT_NAMES)})


class Context:
    """One HIP device (rt_init / rt_shutdown)."""

    def __init__(self, device_id: int = 0):
        self._L = load()
        self._h = C.c_void_p()
        self.device_id = device_id
        _check(self._L.rt_init(device_id, C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.rt_shutdown(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self) -> dict:
        out = abi.rt_device_info()
        _check(self._L.rt_get_device_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in abi.rt_device_info._fields_}

    def scene(self, flat, bvh_builder: int = abi.RT_BVH_DEFAULT) -> "Scene":
        return Scene(self, flat, bvh_builder=bvh_builder)

    def scene_from_file(self, path: str) -> "Scene":
        """rt_scene_create_from_file: the library reads and validates the scene file itself."""
        return Scene(self, None, path=path)

    def scene_from_bytes(self, data: bytes) -> "Scene":
        """rt_scene_create_from_memory on a scene file held in memory."""
        return Scene(self, None, data=data)

    def tonemap_device(self, d_rgb_sum_ptr: int, n_pixels: int, spp_total: int, d_argb_ptr: int, stream: int = 0):
        _check(self._L.rt_tonemap_device(self._h, d_rgb_sum_ptr, n_pixels, spp_total, d_argb_ptr, stream or None))


def _decode(entry: str, data: bytes) -> np.ndarray:
    L = load()
    f = getattr(L, entry)
    w, h = C.c_int32(0), C.c_int32(0)
    rc = f(data, len(data), C.byref(w), C.byref(h), None, 0)
    if rc != 0:
        raise RtError(rc, L.rt_last_error().decode())
    out = np.empty((h.value, w.value, 3), dtype=np.uint8)
    rc = f(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
    if rc != 0:
        raise RtError(rc, L.rt_last_error().decode())
    return out


def decode_jpeg(data: bytes) -> np.ndarray:
    """(image-texture url)'s decode step, texture.clj:68-81, through the library: the bytes of a JPEG
    file -> (h, w, 3) uint8, top row first, R G B. Host code: needs no GPU."""
    return _decode("rt_image_decode_jpeg", data)


def decode_png(data: bytes) -> np.ndarray:
    """The same for the bytes of a PNG file (rt_image_decode_png)."""
    return _decode("rt_image_decode_png", data)


def decode_gif(data: bytes) -> np.ndarray:
    """The first image of a GIF file (rt_image_decode_gif)."""
    return _decode("rt_image_decode_gif", data)


def decode_bmp(data: bytes) -> np.ndarray:
    """A BMP file (rt_image_decode_bmp)."""
    return _decode("rt_image_decode_bmp", data)


def decode_image(data: bytes) -> np.ndarray:
    """ImageIO/read of texture.clj:72: JPEG, PNG, GIF or BMP, told apart by the first bytes (rt_image_decode)."""
    return _decode("rt_image_decode", data)


class MultiContext:
    """Several GPUs behind one handle for a single-process caller (rt_multi_*): one host thread per
    device, one RCCL reduce onto the first device."""

    def __init__(self, device_ids):
        self._L = load()
        self._h = C.c_void_p()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        self._check(self._L.rt_multi_init(ids, len(device_ids), C.byref(self._h)))

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._L.rt_multi_last_error().decode("utf-8", "replace"))

    def close(self):
        if self._h:
            self._L.rt_multi_shutdown(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self) -> int:
        return int(self._L.rt_multi_device_count(self._h))

    def scene(self, flat) -> "MultiScene":
        return MultiScene(self, flat)


class MultiScene:
    def __init__(self, mctx: MultiContext, flat):
        self.mctx = mctx
        self.flat = flat
        self._h = C.c_void_p()
        mctx._check(mctx._L.rt_multi_scene_create(mctx._h, C.byref(flat.desc), C.byref(self._h)))

    def close(self):
        if self._h:
            self.mctx._L.rt_multi_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, params, split=abi.RT_SPLIT_TILES):
        """-> (rgb_sum (H, W, 3) float32, argb (H, W) uint32, stats dict), host buffers as Scene.render."""
        W, H = params.width, params.height
        rgb = np.zeros((H, W, 3), dtype=np.float32)
        argb = np.zeros((H, W), dtype=np.uint32)
        st = abi.rt_stats()
        self.mctx._check(self.mctx._L.rt_multi_render(self.mctx._h, self._h, C.byref(cam), C.byref(params), int(split),
                                                      _fp(rgb), argb.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(st)))
        return rgb, argb, stats_dict(st)


class Scene:
    """An uploaded world (rt_scene_create / rt_scene_destroy)."""

    def __init__(self, ctx: Context, flat, path: str = None, data: bytes = None,
                 bvh_builder: int = abi.RT_BVH_DEFAULT):
        self.ctx = ctx
        self.flat = flat
        self._L = ctx._L
        self._h = C.c_void_p()
        if path is not None:
            _check(self._L.rt_scene_create_from_file(ctx._h, path.encode(), C.byref(self._h)))
        elif data is not None:
            buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
            _check(self._L.rt_scene_create_from_memory(ctx._h, buf, len(data), C.byref(self._h)))
        elif bvh_builder != abi.RT_BVH_DEFAULT:
            _check(self._L.rt_scene_create_with(ctx._h, C.byref(flat.desc), bvh_builder, C.byref(self._h)))
        else:
            _check(self._L.rt_scene_create(ctx._h, C.byref(flat.desc), C.byref(self._h)))

    def close(self):
        if self._h:
            self._L.rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        out = abi.rt_scene_info()
        _check(self._L.rt_scene_get_info(self._h, C.byref(out)))
        return {n: getattr(out, n) for n, _ in abi.rt_scene_info._fields_}

    def render(self, cam: abi.rt_camera, params: abi.rt_params, want_rgb=True, want_argb=True, want_stats=True):
        """Host-buffer render. -> (rgb_sum (H, W, 3) float32 with row 0 = bottom | None,
        argb (H, W) uint32 | None, stats dict | None)."""
        W, H = params.width, params.height
        rgb = np.zeros((H, W, 3), dtype=np.float32) if want_rgb else None
        argb = np.zeros((H, W), dtype=np.uint32) if want_argb else None
        st = abi.rt_stats() if want_stats else None
        _check(self._L.rt_render(self.ctx._h, self._h, C.byref(cam), C.byref(params), _fp(rgb),
                                 argb.ctypes.data_as(C.POINTER(C.c_uint32)) if want_argb else None,
                                 C.byref(st) if want_stats else None))
        return rgb, argb, (stats_dict(st) if want_stats else None)

    def render_device(self, cam, params, d_rgb_sum_ptr: int = 0, d_argb_ptr: int = 0, want_stats=False, stream: int = 0):
        """Device-buffer render into raw device pointers (e.g. torch tensors' data_ptr())."""
        st = abi.rt_stats() if want_stats else None
        _check(self._L.rt_render_device(self.ctx._h, self._h, C.byref(cam), C.byref(params),
                                        d_rgb_sum_ptr or None, d_argb_ptr or None,
                                        C.byref(st) if want_stats else None, stream or None))
        return stats_dict(st) if want_stats else None

    def render_device_noalloc(self, cam, params, d_rgb_sum_ptr: int = 0, d_argb_ptr: int = 0, stream: int = 0):
        """rt_render_device_noalloc: the same render with its scratch in the library's fixed device
        array; allocates nothing, returns no statistics, refuses requests beyond the array's size."""
        _check(self._L.rt_render_device_noalloc(self.ctx._h, self._h, C.byref(cam), C.byref(params),
                                                d_rgb_sum_ptr or None, d_argb_ptr or None, stream or None))

    def bvh_digest(self):
        """-> (64-bit digest of the tree records, number of records, height of the tallest tree)."""
        d, n, h = C.c_uint64(0), C.c_int32(0), C.c_int32(0)
        _check(self._L.rt_scene_bvh_digest(self._h, C.byref(d), C.byref(n), C.byref(h)))
        return d.value, n.value, h.value

    def variants(self):
        """-> (instantiation the render uses, list of all instantiations that cover the world)."""
        rv, mask = C.c_int32(0), C.c_uint32(0)
        _check(self._L.rt_scene_get_variants(self._h, C.byref(rv), C.byref(mask)))
        return rv.value, [v for v in range(abi.RT_N_VARIANTS) if mask.value >> v & 1]

    def scatter_batch(self, seed, orig, dirs, time, px, smp, bounce):
        """One ray-color invocation per ray with the keyed draws of (px, smp, bounce)
        -> structured array of rt_scatter records."""
        orig = np.ascontiguousarray(orig, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = orig.shape[0]
        tm = None if time is None else np.ascontiguousarray(time, dtype=np.float32)
        ia = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int32), (n,))) for a in (px, smp, bounce)]
        out = (abi.rt_scatter * max(n, 1))()
        ipt = C.POINTER(C.c_int32)
        _check(self._L.rt_scatter_batch(self.ctx._h, self._h, int(seed), n, _fp(orig), _fp(dirs), _fp(tm),
                                        ia[0].ctypes.data_as(ipt), ia[1].ctypes.data_as(ipt),
                                        ia[2].ctypes.data_as(ipt), out))
        dt = np.dtype([("hit", "i4"), ("ok", "i4"), ("mat_id", "i4"), ("pad", "i4"), ("orig", "f4", 3),
                       ("dir", "f4", 3), ("time", "f4"), ("attenuation", "f4", 3), ("emitted", "f4", 3)])
        return np.frombuffer(out, dtype=dt, count=n).copy()

    def hit_batch(self, orig, dirs, time=None, t_min=0.001, t_max=float("inf"), medium_xi=None, variant=None):
        """-> structured array of rt_hit records. variant: None = rt_hit_batch, else the kernel
        instantiation for rt_hit_batch_variant (abi.RT_VARIANT_RENDER, 0..5)."""
        orig = np.ascontiguousarray(orig, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        n = orig.shape[0]
        tm = None if time is None else np.ascontiguousarray(time, dtype=np.float32)
        xi = None if medium_xi is None else np.ascontiguousarray(medium_xi, dtype=np.float32)
        out = (abi.rt_hit * max(n, 1))()
        if variant is None:
            _check(self._L.rt_hit_batch(self.ctx._h, self._h, n, _fp(orig), _fp(dirs), _fp(tm), t_min, t_max,
                                        _fp(xi), out))
        else:
            _check(self._L.rt_hit_batch_variant(self.ctx._h, self._h, int(variant), n, _fp(orig), _fp(dirs),
                                                _fp(tm), t_min, t_max, _fp(xi), out))
        dt = np.dtype([("t", "f4"), ("p", "f4", 3), ("normal", "f4", 3), ("u", "f4"), ("v", "f4"),
                       ("front_face", "i4"), ("mat_id", "i4"), ("hit", "i4")])
        return np.frombuffer(out, dtype=dt, count=n).copy()

    def sample_batch(self, cam, params, px, smp):
        px = np.ascontiguousarray(px, dtype=np.int32)
        smp = np.ascontiguousarray(smp, dtype=np.int32)
        out = np.zeros((px.shape[0], 3), dtype=np.float32)
        _check(self._L.rt_sample_batch(self.ctx._h, self._h, C.byref(cam), C.byref(params), px.shape[0],
                                       px.ctypes.data_as(C.POINTER(C.c_int32)),
                                       smp.ctypes.data_as(C.POINTER(C.c_int32)), _fp(out)))
        return out
