"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (raytracing_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp, fp, ip = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        descp = C.POINTER(abi.rt_scene_desc)
        L.orc_hit_batch.argtypes = [descp, C.c_int32, fp, fp, fp, C.c_double, C.c_double, fp, C.c_int32, dp]
        L.orc_render.argtypes = [descp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), dp,
                                 C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint64)]
        L.orc_sample_batch.argtypes = [descp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params),
                                       C.c_int32, ip, ip, dp]
        L.orc_path_rays.argtypes = [descp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), C.c_int32,
                                    C.c_int32, C.c_int32, dp]
        L.orc_pack_pixel.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int32]
        L.orc_pack_pixel.restype = C.c_uint32
        L.orc_tonemap_f32.argtypes = [fp, C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]
        L.orc_camera.argtypes = [dp, dp, dp] + [C.c_double] * 6 + [C.POINTER(abi.rt_camera)]
        L.orc_perlin_noise.argtypes = [descp, C.c_int32, dp]
        L.orc_perlin_noise.restype = C.c_double
        L.orc_perlin_turb.argtypes = [descp, C.c_int32, dp, C.c_int32]
        L.orc_perlin_turb.restype = C.c_double
        L.orc_texture_value.argtypes = [descp, C.c_int32, C.c_double, C.c_double, dp, dp]
        L.orc_sampler_scripted.argtypes = [C.c_int32, dp, C.c_int32, dp]
        L.orc_sampler_keyed.argtypes = [C.c_int32, C.c_uint64, C.c_int32, dp]
        L.orc_scatter_scripted.argtypes = [descp, C.c_int32, dp, dp, dp, C.c_int32, dp]
        L.orc_sample_scripted.argtypes = [descp, C.POINTER(abi.rt_camera), C.POINTER(abi.rt_params), C.c_int32, C.c_int32,
                                          dp, C.c_int32, dp]
        L.orc_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.orc_reflectance.argtypes = [C.c_double, C.c_double]
        L.orc_reflectance.restype = C.c_double
        L.orc_refract.argtypes = [dp, dp, C.c_double, dp]
        L.orc_aabb_hit.argtypes = [dp, dp, dp, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


HIT_FIELDS = ("t", "px", "py", "pz", "nx", "ny", "nz", "u", "v", "front_face", "mat_id", "hit")


def hit_batch(flat, orig, dirs, time=None, t_min=0.001, t_max=float("inf"), medium_xi=None):
    """-> (n, 12) float64: t p[3] normal[3] u v front_face mat_id hit."""
    orig = np.ascontiguousarray(orig, dtype=np.float32)
    dirs = np.ascontiguousarray(dirs, dtype=np.float32)
    n = orig.shape[0]
    tm = None if time is None else np.ascontiguousarray(time, dtype=np.float32)
    xi = None if medium_xi is None else np.ascontiguousarray(medium_xi, dtype=np.float32)
    out = np.zeros((n, 12), dtype=np.float64)
    rc = lib().orc_hit_batch(C.byref(flat.desc), n, _fp(orig), _fp(dirs), _fp(tm), t_min, t_max,
                             _fp(xi), 0 if xi is not None else 1, _dp(out))
    assert rc == 0
    return out


def render(flat, cam, params, nthreads=1, want_argb=True):
    """-> (rgb_sum (H, W, 3) float64 with row 0 = bottom, argb (H, W) uint32 or None, rays)."""
    W, H = params.width, params.height
    rgb = np.zeros((H, W, 3), dtype=np.float64)
    argb = np.zeros((H, W), dtype=np.uint32) if want_argb else None
    rays = C.c_uint64(0)
    rc = lib().orc_render(C.byref(flat.desc), C.byref(cam), C.byref(params), _dp(rgb),
                          argb.ctypes.data_as(C.POINTER(C.c_uint32)) if want_argb else None,
                          int(nthreads), C.byref(rays))
    assert rc == 0
    return rgb, argb, rays.value


def sample_batch(flat, cam, params, px, smp):
    px = np.ascontiguousarray(px, dtype=np.int32)
    smp = np.ascontiguousarray(smp, dtype=np.int32)
    out = np.zeros((px.shape[0], 3), dtype=np.float64)
    rc = lib().orc_sample_batch(C.byref(flat.desc), C.byref(cam), C.byref(params), px.shape[0],
                                px.ctypes.data_as(C.POINTER(C.c_int32)),
                                smp.ctypes.data_as(C.POINTER(C.c_int32)), _dp(out))
    assert rc == 0
    return out


def path_rays(flat, cam, params, px, smp, max_rays=64):
    """Rays one (pixel, sample) path hands to `hit`: (n, 7) float64 orig dir time."""
    out = np.zeros((max_rays, 7), dtype=np.float64)
    n = lib().orc_path_rays(C.byref(flat.desc), C.byref(cam), C.byref(params), int(px), int(smp), max_rays, _dp(out))
    return out[:min(n, max_rays)]


def camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus, time0, time1) -> abi.rt_camera:
    out = abi.rt_camera()
    a = [np.asarray(v, dtype=np.float64) for v in (lookfrom, lookat, vup)]
    lib().orc_camera(_dp(a[0]), _dp(a[1]), _dp(a[2]), vfov, aspect, aperture, focus, time0, time1,
                     C.byref(out))
    return out


def pack_pixel(r, g, b, spp) -> int:
    return int(lib().orc_pack_pixel(r, g, b, spp))


def tonemap_f32(rgb_sum, spp):
    rgb = np.ascontiguousarray(rgb_sum, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(rgb.shape[0], dtype=np.uint32)
    lib().orc_tonemap_f32(_fp(rgb), rgb.shape[0], spp, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def perlin_noise(flat, pid, p) -> float:
    return lib().orc_perlin_noise(C.byref(flat.desc), pid, _dp(np.asarray(p, dtype=np.float64)))


def perlin_turb(flat, pid, p, depth=7) -> float:
    return lib().orc_perlin_turb(C.byref(flat.desc), pid, _dp(np.asarray(p, dtype=np.float64)), depth)


def texture_value(flat, tex, u, v, p):
    out = np.zeros(3)
    lib().orc_texture_value(C.byref(flat.desc), tex, u, v, _dp(np.asarray(p, dtype=np.float64)), _dp(out))
    return out


def sampler_scripted(which, script):
    s = np.asarray(script, dtype=np.float64)
    out = np.zeros(3)
    used = lib().orc_sampler_scripted(which, _dp(s), s.shape[0], _dp(out))
    return out, used


def sampler_keyed(which, seed, n):
    out = np.zeros((n, 3))
    lib().orc_sampler_keyed(which, seed, n, _dp(out))
    return out


def scatter_scripted(flat, mat, ray7, rec9, script):
    s = np.asarray(script if len(script) else [0.5], dtype=np.float64)
    out = np.zeros(14)
    used = lib().orc_scatter_scripted(C.byref(flat.desc), mat, _dp(np.asarray(ray7, dtype=np.float64)),
                                      _dp(np.asarray(rec9, dtype=np.float64)), _dp(s), len(script), _dp(out))
    return dict(ok=bool(out[0]), attenuation=out[1:4], orig=out[4:7], dir=out[7:10], time=out[10],
                emitted=out[11:14], used=used)


def sample_scripted(flat, cam, params, x, y, script):
    """One sample of pixel (x, y) with scripted uniforms in the reference's evaluation order -> (rgb, uniforms consumed)."""
    sc = np.asarray(script, dtype=np.float64)
    out = np.zeros(3)
    used = lib().orc_sample_scripted(C.byref(flat.desc), C.byref(cam), C.byref(params), int(x), int(y), _dp(sc), sc.shape[0], _dp(out))
    return out, used


def philox(ctr, key, rounds=0):
    """Philox-4x32 with `rounds` rounds; 0 = the round count of the project's keyed stream."""
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox(c, k, rounds, o)
    return list(o)
