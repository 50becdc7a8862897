"""ctypes view of include/rtmi.h: structures and enum values, nothing else."""
from __future__ import annotations

import ctypes as C

RTMI_ABI_VERSION = 2

RT_OK, RT_ERR_ARG, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_OOM = 0, -1, -2, -3, -4

RT_PRIM_SPHERE, RT_PRIM_MOVING_SPHERE, RT_PRIM_XY_RECT, RT_PRIM_XZ_RECT, RT_PRIM_YZ_RECT = range(5)
RT_PRIM_STRIDE = 12
(RT_NODE_PRIM, RT_NODE_LIST, RT_NODE_BVH, RT_NODE_TRANSLATE, RT_NODE_ROTATE_Y,
 RT_NODE_MEDIUM) = range(6)
RT_NODE_ISTRIDE, RT_NODE_DSTRIDE = 4, 6
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = range(5)
RT_MAT_DSTRIDE = 4
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_NOISE, RT_TEX_IMAGE = range(4)
RT_TEX_DSTRIDE, RT_TEX_ISTRIDE = 4, 2
RT_FLAG_COUNTERS = 1
RT_FLAG_POOL_PATHS = 2
RT_SPLIT_TILES, RT_SPLIT_SAMPLES = 0, 1
RT_BVH_DEFAULT, RT_BVH_DEVICE_SAH, RT_BVH_HOST_SAH, RT_BVH_DEVICE_LBVH = 0, 1, 2, 3
RT_N_VARIANTS, RT_VARIANT_GENERAL, RT_VARIANT_RENDER = 8, -1, -2
VARIANT_NAMES = ["V_RECT", "V_SMOKE", "V_LIST", "V_SPH", "V_NOISE", "V_ALL_DEEP", "V_FULL", "V_BALLS"]

RT_EVENT_NAMES = [
    "aabb_test", "sphere_test", "sphere_root", "sphere_record", "msphere_center", "rect_test",
    "rect_record", "rotate_y_ray", "rotate_y_rec", "translate_ray", "translate_rec",
    "medium_sample", "camera_ray", "unit_sphere_trial", "unit_disk_trial", "lambert_scatter",
    "metal_scatter", "dielectric_scatter", "iso_scatter", "light_emit", "checker", "perlin_noise",
    "noise_tex", "image_tex", "throughput_mul", "accumulate", "box_slab",
]
RT_EV_COUNT = len(RT_EVENT_NAMES)

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class rt_scene_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_prims", C.c_int32), ("prim_type", _i32p), ("prim_d", _f64p), ("prim_mat", _i32p),
        ("n_nodes", C.c_int32), ("node_kind", _i32p), ("node_i", _i32p), ("node_d", _f64p),
        ("n_children", C.c_int32), ("node_children", _i32p),
        ("root", C.c_int32), ("n_media", C.c_int32),
        ("n_mats", C.c_int32), ("mat_type", _i32p), ("mat_d", _f64p), ("mat_tex", _i32p),
        ("n_tex", C.c_int32), ("tex_type", _i32p), ("tex_d", _f64p), ("tex_i", _i32p),
        ("n_perlin", C.c_int32), ("perlin_ranvec", _f64p), ("perlin_perm", _i32p),
        ("n_images", C.c_int32), ("img_w", _i32p), ("img_h", _i32p), ("img_rgb8", C.POINTER(_u8p)),
    ]


class rt_camera(C.Structure):
    _fields_ = [(n, C.c_double * 3) for n in
                ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w")] + \
               [("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32),
                ("sample_offset", C.c_int32), ("max_depth", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_int32),
                ("background", C.c_double * 3), ("seed", C.c_uint64)]


class rt_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64),
                ("events", C.c_uint64 * RT_EV_COUNT), ("kernel_ms", C.c_double),
                ("shader_mhz", C.c_double)]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_float), ("p", C.c_float * 3), ("normal", C.c_float * 3),
                ("u", C.c_float), ("v", C.c_float),
                ("front_face", C.c_int32), ("mat_id", C.c_int32), ("hit", C.c_int32)]


class rt_scatter(C.Structure):
    _fields_ = [("hit", C.c_int32), ("ok", C.c_int32), ("mat_id", C.c_int32), ("pad", C.c_int32),
                ("orig", C.c_float * 3), ("dir", C.c_float * 3), ("time", C.c_float),
                ("attenuation", C.c_float * 3), ("emitted", C.c_float * 3)]


class rt_scene_info(C.Structure):
    _fields_ = [("n_prims_flat", C.c_int32), ("n_groups", C.c_int32), ("n_chains", C.c_int32),
                ("n_media", C.c_int32), ("n_bvh_nodes", C.c_int32), ("lds_bytes", C.c_int32)]


class rt_device_info(C.Structure):
    _fields_ = [("n_cu", C.c_int32), ("clock_khz", C.c_int32), ("lds_bytes_max", C.c_int32), ("timer_khz", C.c_int32)]


def make_camera(cam) -> rt_camera:
    """camera.Camera -> rt_camera."""
    out = rt_camera()
    for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        getattr(out, name)[:] = getattr(cam, name)
    out.lens_radius, out.time0, out.time1 = cam.lens_radius, cam.time0, cam.time1
    return out


def make_params(width, height, spp, max_depth=50, background=(0.0, 0.0, 0.0), seed=0x5EED0001,
                sample_offset=0, shard_index=0, shard_count=1, flags=0) -> rt_params:
    p = rt_params()
    p.width, p.height, p.spp, p.sample_offset = int(width), int(height), int(spp), int(sample_offset)
    p.max_depth, p.shard_index, p.shard_count, p.flags = int(max_depth), int(shard_index), int(shard_count), int(flags)
    p.background[:] = [float(x) for x in background]
    p.seed = int(seed)
    return p
