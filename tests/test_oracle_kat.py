"""Known-answer tests that pin the CPU oracle (oracle/oracle.cpp).

The reference ships no tests (test/raytracing/core_test.clj:1-3 is an empty namespace), so every
expected value here is derived by hand from the cited reference lines (paths relative to
/root/reference/src/raytracing/). PARITY UNPINNED by the reference itself.
"""
import math

import numpy as np
import pytest

from raytracing_amd import abi, flatten
from raytracing_amd.hittables import (box, constant_medium, rotate_y, sphere, translate, xy_rect,
                                      xz_rect, yz_rect, moving_sphere)
from raytracing_amd.materials import Metal, dielectric, diffuse_light, lambertian, metal
from raytracing_amd.texture import (Perlin, NoiseTexture, checker_texture, image_texture,
                                    noise_texture, solid_color)

GREY = lambertian([0.5, 0.5, 0.5])


def hit1(oracle, world, o, d, tm=0.0, t_min=0.001, t_max=float("inf"), xi=None):
    flat = flatten.flatten_world(world)
    r = oracle.hit_batch(flat, [o], [d], [tm], t_min, t_max, None if xi is None else [xi])[0]
    return dict(zip(oracle.HIT_FIELDS, r))


def test_k1_sphere_front(oracle):  # base.clj:234-246,222-227
    h = hit1(oracle, [sphere([0, 0, 0], 1, GREY)], [0, 0, -5], [0, 0, 1])
    assert h["hit"] == 1 and h["t"] == 4.0
    assert (h["px"], h["py"], h["pz"]) == (0.0, 0.0, -1.0)
    assert (h["nx"], h["ny"], h["nz"]) == (0.0, 0.0, -1.0)
    assert h["front_face"] == 1
    assert h["u"] == pytest.approx(0.75, abs=1e-15) and h["v"] == pytest.approx(0.5, abs=1e-15)


def test_k2_inside_second_root_misses(oracle):  # base.clj:247-248: root2 = (-hb + root1)/a = -1
    h = hit1(oracle, [sphere([0, 0, 0], 1, GREY)], [0, 0, 0], [0, 0, 1])
    assert h["hit"] == 0


def test_k3_inside_second_root_hits(oracle):
    # o = [0 0 -0.875] (exact in FP32): hb = -0.875, c = -0.234375, disc = 1; root1 = -0.125
    # rejected; root2 = (0.875 - 0.125)/1 = 0.75 -> p.z = -0.125, outward normal [0 0 -0.125]
    h = hit1(oracle, [sphere([0, 0, 0], 1, GREY)], [0, 0, -0.875], [0, 0, 1])
    assert h["hit"] == 1 and h["t"] == 0.75
    assert h["pz"] == -0.125 and h["nz"] == -0.125 and h["front_face"] == 1
    assert h["u"] == pytest.approx(0.75) and h["v"] == pytest.approx(0.5)


def test_k4_second_root_depends_on_direction_length(oracle):
    # same ray with d = [0 0 2]: a = 4, hb = -1.75, disc = 3.0625 + 0.9375 = 4, root1 = -0.0625,
    # root2 = (1.75 - 0.0625)/4 = 0.421875 -> p.z = -0.03125 (K3 gave -0.125)
    h = hit1(oracle, [sphere([0, 0, 0], 1, GREY)], [0, 0, -0.875], [0, 0, 2])
    assert h["t"] == 0.421875 and h["pz"] == -0.03125


def test_k5_xy_rect_back_face(oracle):  # rectangles.clj:20-37
    h = hit1(oracle, [xy_rect(0, 555, 0, 555, 555, GREY)], [278, 278, -800], [0, 0, 1])
    assert h["t"] == 1355.0 and h["u"] == 278 / 555 and h["v"] == 278 / 555
    assert h["front_face"] == 0 and h["nz"] == -1.0 and h["nx"] == 0.0 and h["ny"] == 0.0


def test_rect_bounds_and_t_range_are_inclusive(oracle):  # rectangles.clj:23-30
    r = [xy_rect(0, 1, 0, 1, 2, GREY)]
    assert hit1(oracle, r, [1, 1, 0], [0, 0, 1])["hit"] == 1          # corner, inclusive
    assert hit1(oracle, r, [1.0000001, 1, 0], [0, 0, 1])["hit"] == 0
    assert hit1(oracle, r, [0.5, 0.5, 0], [0, 0, 1], t_min=2.0, t_max=2.0)["hit"] == 1  # t == both limits
    assert hit1(oracle, r, [0.5, 0.5, 0], [0, 0, 1], t_max=1.9999)["hit"] == 0


def test_sphere_t_range_is_open(oracle):  # base.clj:242: (> t-max root t-min)
    s = [sphere([0, 0, 0], 1, GREY)]
    assert hit1(oracle, s, [0, 0, -5], [0, 0, 1], t_max=4.0)["hit"] == 0
    assert hit1(oracle, s, [0, 0, -5], [0, 0, 1], t_max=4.0000001)["hit"] == 1
    # tangent ray: discriminant exactly 0 is a miss ((pos? 0) is false), base.clj:239
    assert hit1(oracle, s, [1, 0, -5], [0, 0, 1])["hit"] == 0


def test_k6_rect_k_is_truncated_to_long():  # rectangles.clj:50,75-77
    r = xz_rect(0, 1, 0, 1, 7.9, GREY)
    assert r.k == 7.0
    bb = r.bounding_box(0, 1)
    assert bb.minimum[1] == 7 - 0.0001 and bb.maximum[1] == 7 + 0.0001
    assert yz_rect(0, 1, 0, 1, -7.9, GREY).k == -7.0  # toward zero


def test_k7_rotate_y_uses_plus_sin_on_the_way_back(oracle):  # rotate.clj:18-47
    w = [rotate_y(xy_rect(-1, 1, -1, 1, 0, GREY), 30)]
    h = hit1(oracle, w, [0.5, 0, -5], [0, 0, 1])
    s, c = math.sin(math.radians(30)), math.cos(math.radians(30))
    # rotated ray: o' = [c*.5 + s*5, 0, s*.5 - c*5], d' = [-s, 0, c]; child t = -o'_z / c
    t = -(s * 0.5 - c * 5) / c
    assert h["t"] == pytest.approx(t, rel=1e-15) and t == pytest.approx(4.7113249, abs=1e-7)
    px_child = (c * 0.5 + s * 5) + t * (-s)
    assert px_child == pytest.approx(0.5773503, abs=1e-7)
    assert h["px"] == pytest.approx(c * px_child, rel=1e-14)           # 0.5
    assert h["pz"] == pytest.approx(s * px_child, rel=1e-14)           # +0.2886751, not -0.2886751
    assert h["pz"] > 0
    # child normal: d'.z > 0 -> back face -> [-0 -0 -1]; rotated with +sin: [c*0 + s*(-1), 0, s*0 + c*(-1)]
    assert (h["nx"], h["nz"]) == (pytest.approx(-s), pytest.approx(-c))
    assert h["front_face"] == 1  # d' . n = s*s - c*c = -0.5 < 0: re-derived against the ROTATED ray
    assert h["u"] == pytest.approx((px_child + 1) / 2) and h["v"] == 0.5


def test_k8_translate_rotate_box_matches_an_independent_walk(oracle):
    """Cornell block 1 (scenes.clj:82-84) hit by the camera centre ray, re-derived with numpy."""
    white = lambertian([0.73, 0.73, 0.73])
    blk = translate(rotate_y(box([0, 0, 0], [165, 330, 165], white), 15), [265, 0, 295])
    o = np.array([278.0, 278.0, -800.0]); d = np.array([0.0625, -0.03125, 1.0])  # exact in FP32
    h = hit1(oracle, [blk], o, d)
    from util import cornell_block_walk
    t, p3, n2, front = cornell_block_walk(o, d)
    assert h["hit"] == 1 and h["t"] == pytest.approx(t, rel=1e-14)
    assert [h["px"], h["py"], h["pz"]] == pytest.approx(list(p3), rel=1e-13)
    assert [h["nx"], h["ny"], h["nz"]] == pytest.approx(list(n2), abs=1e-15)
    assert h["front_face"] == int(front)


def test_k9_constant_medium_with_sphere_boundary(oracle):  # constant_medium.clj:12-44 + base.clj:247
    m = constant_medium(sphere([0, 0, 0], 1, GREY), 1.0, [1, 1, 1])
    h = hit1(oracle, [m], [0, 0, -5], [0, 0, 1], xi=[math.exp(-1)])
    # rec1.t = 4; rec2: root1 = 4 is not > 4.0001, root2 = (5 + 4)/1 = 9 (the true exit is 6);
    # inside = 9 - 4 = 5, hit_distance = -1 * ln(e^-1) = 1 -> t = 5
    assert h["hit"] == 1 and h["t"] == pytest.approx(5.0, rel=1e-7)   # xi is FP32
    assert (h["px"], h["py"]) == (0, 0) and h["pz"] == pytest.approx(0.0, abs=1e-6)
    assert (h["nx"], h["ny"], h["nz"]) == (1.0, 0.0, 0.0) and h["front_face"] == 1
    # hit_distance 5.0000001 > inside: no hit
    assert hit1(oracle, [m], [0, 0, -5], [0, 0, 1], xi=[math.exp(-5.001)])["hit"] == 0
    # from inside the sphere rec2.t < rec1.t' ... (second root again): no hit at all from the centre
    assert hit1(oracle, [m], [0, 0, 0], [0, 0, 1], xi=[0.9])["hit"] == 0


def _perlin_const(vec=(1.0, 0.0, 0.0)):
    ident = np.arange(256, dtype=np.int32)
    return Perlin(256, np.tile(np.array(vec), (256, 1)), ident.copy(), ident.copy(), ident.copy())


def test_k10_perlin_double_hermite(oracle):  # noise.clj:21-33,39-58
    tex = NoiseTexture(_perlin_const(), 4)
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(tex))])
    assert oracle.perlin_noise(flat, 0, [0.5, 0.5, 0.5]) == pytest.approx(0.0, abs=1e-17)
    u = 0.25 * 0.25 * (3 - 0.5)          # 0.15625
    uu = u * u * (3 - 2 * u)             # 0.06561279296875
    assert oracle.perlin_noise(flat, 0, [0.25, 0.25, 0.25]) == pytest.approx(u - uu, rel=1e-14)
    assert u - uu == pytest.approx(0.09063720703125)
    # negative coordinates: floor, not truncation (noise.clj:41,46)
    assert oracle.perlin_noise(flat, 0, [-0.75, 0.25, 0.25]) == pytest.approx(u - uu, rel=1e-14)


def test_perlin_gradient_index_is_bit_or(oracle):  # noise.clj:53-55
    ranvec = np.zeros((256, 3)); ranvec[:, 0] = np.arange(256)  # gradient x = its own index
    ident = np.arange(256, dtype=np.int32)
    tex = NoiseTexture(Perlin(256, ranvec, ident.copy(), ident.copy(), ident.copy()), 1)
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(tex))])
    # p = (1, 2, 4) exactly: u = v = w = 0 -> only corner (0,0,0) has weight 1... times dot(c, [0,0,0]) = 0.
    # Use p = (1.5, 2, 4): u = h(.5) = .5, uu = .5; v = w = 0 -> corners (i,0,0):
    # noise = (1-uu) * idx(1|2|4) * (u-0) + uu * idx(2|2|4) * (u-1) = .5*7*.5 + .5*6*(-.5) = 0.25
    assert oracle.perlin_noise(flat, 0, [1.5, 2.0, 4.0]) == pytest.approx(0.25, rel=1e-14)  # xor would give .5*7*.5 - .5*4*.5


def test_turbulence_and_marble(oracle):  # noise.clj:59-70, texture.clj:41-44
    tex = NoiseTexture(_perlin_const(), 4)
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(tex))])
    p = np.array([0.3, 0.7, 0.2])
    acc, w, tp = 0.0, 1.0, p.copy()
    for _ in range(7):
        acc += w * oracle.perlin_noise(flat, 0, tp)
        w *= 0.5
        tp = tp * 2
    assert oracle.perlin_turb(flat, 0, p, 7) == pytest.approx(abs(acc), rel=1e-15)
    val = oracle.texture_value(flat, 0, 0.0, 0.0, p)
    k = 0.5 * (1 + math.sin(4 * p[2] + 10 * abs(acc)))
    assert list(val) == pytest.approx([k, k, k], rel=1e-14)


def test_k11_noise_scale_is_truncated_to_long():  # texture.clj:38-39,48
    import random
    assert noise_texture(random.Random(0), 0.1).scale == 0
    assert noise_texture(random.Random(0), 4.0).scale == 4
    assert noise_texture(random.Random(0)).scale == 1


def test_k12_unit_sphere_sampler_keeps_the_outside(oracle):  # base.clj:182-186
    v = oracle.sampler_keyed(0, 1234, 200000)
    l2 = (v * v).sum(axis=1)
    assert l2.min() >= 1.0 and l2.max() <= 3.0
    # the region [-1,1]^3 minus the ball has volume 8 - 4pi/3; mean |x| there is analytic-free, so
    # check the support and the octant symmetry only
    assert abs((v > 0).mean() - 0.5) < 0.01
    # disk sampler (base.clj:188-194) keeps the inside; z = 0
    d = oracle.sampler_keyed(1, 1234, 200000)
    assert ((d[:, :2] ** 2).sum(axis=1) < 1.0).all() and (d[:, 2] == 0).all()
    # uniform on the disk: E[r^2] = 1/2
    assert abs((d[:, :2] ** 2).sum(axis=1).mean() - 0.5) < 0.01
    u = oracle.sampler_keyed(2, 99, 1000)
    assert np.allclose((u * u).sum(axis=1), 1.0, atol=1e-12)


def test_sampler_draw_order(oracle):
    # draw 0 seeds the reduce and is never tested, draw 1 lies inside the ball and is skipped,
    # draw 2 lies outside and is returned: 9 uniforms consumed (base.clj:182-186, :176-180)
    script = [0.5, 0.5, 0.5,   0.6, 0.6, 0.6,   0.95, 0.9, 0.05,   0.1, 0.1, 0.1]
    v, used = oracle.sampler_scripted(0, script)
    assert used == 9 and list(v) == pytest.approx([0.9, 0.8, -0.9])
    # draw 0 outside the ball is still not returned
    v, used = oracle.sampler_scripted(0, [1.0, 1.0, 1.0,   0.0, 0.0, 1.0])
    assert used == 6 and list(v) == [-1.0, -1.0, 1.0]
    # disk: draw 0 skipped, first later draw inside is returned (base.clj:188-194)
    v, used = oracle.sampler_scripted(1, [0.5, 0.5,   1.0, 1.0,   0.75, 0.25])
    assert used == 6 and list(v) == [0.5, -0.5, 0.0]


def test_k13_pixel_pack(oracle):  # core.clj:106-108 + java.awt.Color(float,float,float)
    assert oracle.pack_pixel(0.25, 1.5, float("nan"), 1) == 0xFF80FF00
    assert oracle.pack_pixel(4 * 0.25, 4 * 1.5, 0.0, 4) == 0xFF80FF00       # 1/spp scale
    assert oracle.pack_pixel(1e9, 1e9, 1e9, 1) == 0xFFFFFFFF                 # clamp 0.999 -> 255
    assert oracle.pack_pixel(0.0, 0.0, 0.0, 7) == 0xFF000000
    assert oracle.pack_pixel(-1.0, 0.0, 0.0, 1) == 0xFF000000                # sqrt(<0) = NaN -> 0
    # (int)(c*255 + 0.5) rounds half up: sqrt(x) = 0.5/255 -> 1
    x = (0.5 / 255) ** 2
    assert oracle.pack_pixel(x * 1.0001, 0, 0, 1) & 0x00FF0000 == 0x00010000
    assert oracle.pack_pixel(x * 0.999, 0, 0, 1) & 0x00FF0000 == 0


def test_k14_image_texel_lookup(oracle):  # texture.clj:55-66
    img = np.zeros((4, 8, 3), dtype=np.uint8)
    img[0, 0] = (10, 20, 30); img[3, 7] = (40, 50, 60); img[1, 2] = (70, 80, 90)
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(image_texture(img)))])
    p = [0.0, 0.0, 0.0]
    assert list(oracle.texture_value(flat, 0, 0.0, 1.0, p)) == pytest.approx([10 / 255, 20 / 255, 30 / 255])
    assert list(oracle.texture_value(flat, 0, 1.0, 0.0, p)) == pytest.approx([40 / 255, 50 / 255, 60 / 255])
    assert list(oracle.texture_value(flat, 0, 2.5 / 8, 1 - 1.5 / 4, p)) == pytest.approx([70 / 255, 80 / 255, 90 / 255])
    assert list(oracle.texture_value(flat, 0, -3.0, 9.0, p)) == pytest.approx([10 / 255, 20 / 255, 30 / 255])  # clamped
    assert list(oracle.texture_value(flat, 0, float("nan"), float("nan"), p)) == pytest.approx([10 / 255, 20 / 255, 30 / 255])  # (int NaN) = 0
    flat0 = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(image_texture()))])
    assert list(oracle.texture_value(flat0, 0, 0.3, 0.3, p)) == [0.0, 1.0, 1.0]  # no data, texture.clj:56-57


def test_checker(oracle):  # texture.clj:25-30
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian(checker_texture([1, 0, 0], [0, 1, 0])))])
    tex = 2  # two solids are interned before the checker that refers to them
    assert flat.tex_type[tex] == abi.RT_TEX_CHECKER
    assert list(oracle.texture_value(flat, tex, 0, 0, [0.1, 0.1, 0.1])) == [0.0, 1.0, 0.0]      # +++ -> even
    assert list(oracle.texture_value(flat, tex, 0, 0, [0.1, 0.1, 0.4])) == [1.0, 0.0, 0.0]      # sin(4) < 0 -> odd
    assert list(oracle.texture_value(flat, tex, 0, 0, [-0.1, 0.1, 0.4])) == [0.0, 1.0, 0.0]     # two negatives
    assert list(oracle.texture_value(flat, tex, 0, 0, [0.0, 0.1, 0.4])) == [0.0, 1.0, 0.0]      # product is -0.0: (neg? -0.0) is false


def test_aabb_slab(oracle):  # base.clj:59-75
    from oracle.binding import lib, _dp
    mn, mx = np.array([0.0, 0, 0]), np.array([1.0, 1, 1])
    def hit(o, d, t0=0.001, t1=np.inf):
        return lib().orc_aabb_hit(_dp(mn), _dp(mx), _dp(np.array(list(o) + list(d), dtype=np.float64)), t0, t1)
    assert hit([0.5, 0.5, -1], [0, 0, 1]) == 1
    assert hit([0.5, 0.5, -1], [0, 0, -1]) == 0
    assert hit([1.5, 0.5, -1], [0, 0, 1]) == 0
    assert hit([0.5, 0.5, -1], [0, 0, 1], 0.001, 1.0) == 0   # t-max <= t-min is a miss: interval [1, 1]
    assert hit([0.5, 0.5, -1], [0, 0, 1], 0.001, 1.0001) == 1


def test_lambertian_scatter_scripted(oracle):  # materials.clj:16-23
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, lambertian([0.1, 0.2, 0.3]))])
    rec = [1, 2, 3, 0, 1, 0, 0.5, 0.5, 1]
    script = [0.5, 0.5, 0.5,   1.0, 0.5, 0.5]      # draw 0 unused; draw 1 = [1 0 0] accepted (|v|^2 = 1)
    s = oracle.scatter_scripted(flat, 0, [0, 5, 0, 0, -1, 0, 0.25], rec, script)
    assert s["ok"] and s["used"] == 6
    assert list(s["dir"]) == [1.0, 1.0, 0.0] and list(s["orig"]) == [1.0, 2.0, 3.0] and s["time"] == 0.25
    assert list(s["attenuation"]) == [0.1, 0.2, 0.3] and list(s["emitted"]) == [0.0, 0.0, 0.0]


def test_metal_scatter_scripted(oracle):  # materials.clj:33-39; `metal` clamps fuzz, ->Metal does not
    assert metal([1, 1, 1], 3.0).fuzz == 1 and Metal([1, 1, 1], 3.0).fuzz == 3.0
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, Metal([0.7, 0.6, 0.5], 0.5))])
    rec = [0, 0, 0, 0, 1, 0, 0, 0, 1]
    script = [0.5] * 3 + [0.5, 1.0, 0.5]           # fuzz vector [0 1 0] (length 1, kept: outside-or-on)
    s = oracle.scatter_scripted(flat, 0, [0, 5, 0, 3, -4, 0, 0.0], rec, script)
    # unit d = [.6 -.8 0]; reflect about n = [0 1 0] -> [.6 .8 0]; + 0.5*[0 1 0]
    assert s["ok"] and list(s["dir"]) == pytest.approx([0.6, 1.3, 0.0], rel=1e-15)
    assert list(s["attenuation"]) == [0.7, 0.6, 0.5]
    # a fuzz vector that pushes the direction below the surface: ok = false
    script = [0.5] * 3 + [0.5, 0.0, 0.5]           # [0 -1 0]
    flat2 = flatten.flatten_world([sphere([0, 0, 0], 1, Metal([0.7, 0.6, 0.5], 1.0))])
    s = oracle.scatter_scripted(flat2, 0, [0, 5, 0, 3, -4, 0, 0.0], rec, script)
    assert not s["ok"]


def test_dielectric_scatter_scripted(oracle):  # materials.clj:47-61, base.clj:199-220
    flat = flatten.flatten_world([sphere([0, 0, 0], 1, dielectric(1.5))])
    rec = [0, 0, 0, 0, 1, 0, 0, 0, 1]  # front face
    # normal incidence: cos = 1, reflectance = r0 = ((1 - 1/1.5)/(1 + 1/1.5))^2 = 0.04
    r0 = ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2
    assert oracle.lib().orc_reflectance(1.0, 1 / 1.5) == pytest.approx(r0, rel=1e-15)
    s = oracle.scatter_scripted(flat, 0, [0, 5, 0, 0, -2, 0, 0.0], rec, [0.5])   # 0.04 > 0.5 false -> refract
    assert s["ok"] and s["used"] == 1 and list(s["dir"]) == pytest.approx([0, -1, 0], abs=1e-15)
    s = oracle.scatter_scripted(flat, 0, [0, 5, 0, 0, -2, 0, 0.0], rec, [0.01])  # 0.04 > 0.01 -> reflect
    assert list(s["dir"]) == pytest.approx([0, 1, 0], abs=1e-15)
    assert list(s["attenuation"]) == [1.0, 1.0, 1.0]
    # total internal reflection from inside: ratio = 1.5, sin = 0.8 -> 1.2 > 1: no uniform is drawn
    rec_in = [0, 0, 0, 0, 1, 0, 0, 0, 0]
    s = oracle.scatter_scripted(flat, 0, [0, 5, 0, 0.8, -0.6, 0, 0.0], rec_in, [])
    assert s["used"] == 0 and list(s["dir"]) == pytest.approx([0.8, 0.6, 0], rel=1e-15)


def test_diffuse_light_emits_and_does_not_scatter(oracle):  # materials.clj:69-78
    flat = flatten.flatten_world([xz_rect(0, 1, 0, 1, 5, diffuse_light([15, 15, 15]))])
    s = oracle.scatter_scripted(flat, 0, [0, 0, 0, 0, 1, 0, 0], [0, 5, 0, 0, -1, 0, 0.5, 0.5, 1],
                                [0.5] * 3 + [1.0, 0.5, 0.5])
    assert not s["ok"] and list(s["emitted"]) == [15.0, 15.0, 15.0] and s["used"] == 6  # burns a direction


def test_moving_sphere_centre(oracle):  # base.clj:298-302
    w = [moving_sphere([0, 0, 0], [4, 0, 0], 0, 1, 1, GREY)]
    assert hit1(oracle, w, [1, 0, -5], [0, 0, 1], tm=0.25)["t"] == 4.0      # centre [1 0 0]
    assert hit1(oracle, w, [1, 0, -5], [0, 0, 1], tm=0.75)["hit"] == 0      # centre [3 0 0]


def test_list_tie_break(oracle):  # base.clj:92-101 with the comparisons of the leaves
    a, b = lambertian([1, 0, 0]), lambertian([0, 1, 0])
    # two coincident spheres: the strict (> t-max root) keeps the earlier one
    h = hit1(oracle, [sphere([0, 0, 0], 1, a), sphere([0, 0, 0], 1, b)], [0, 0, -5], [0, 0, 1])
    assert h["mat_id"] == 0
    # two coincident rects: t > t-max is false at equality, the later one replaces the record
    h = hit1(oracle, [xy_rect(0, 1, 0, 1, 2, a), xy_rect(0, 1, 0, 1, 2, b)], [0.5, 0.5, 0], [0, 0, 1])
    assert h["mat_id"] == 1


def test_camera_map(oracle):  # camera.clj:7-33 with core.clj:75-79,89-91 (Cornell)
    from raytracing_amd.camera import camera
    c = camera([278, 278, -800], [278, 278, 0], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    h = math.tan(math.radians(40.0) / 2)
    assert c.w == (0.0, 0.0, -1.0) and c.u == (-1.0, 0.0, 0.0) and c.v == (0.0, 1.0, 0.0)
    assert c.horizontal == pytest.approx((-20 * h, 0, 0)) and c.vertical == pytest.approx((0, 20 * h, 0))
    assert c.lower_left_corner == pytest.approx((278 + 10 * h, 278 - 10 * h, -790))
    assert c.lens_radius == 0.0
    o = oracle.camera([278, 278, -800], [278, 278, 0], [0, 1, 0], 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    for name in ("origin", "lower_left_corner", "horizontal", "vertical", "u", "v", "w"):
        assert tuple(getattr(o, name)) == getattr(c, name)


def test_philox_known_answers(oracle):
    pi_ctr, pi_key = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]
    # Random123 kat_vectors: philox4x32 with 10 rounds (the library default) ...
    assert oracle.philox([0, 0, 0, 0], [0, 0], 10) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2, 10) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox(pi_ctr, pi_key, 10) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # ... and with 7 rounds, the count the keyed stream of this project uses
    assert oracle.philox([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2, 7) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert oracle.philox(pi_ctr, pi_key, 7) == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    assert oracle.philox(pi_ctr, pi_key) == oracle.philox(pi_ctr, pi_key, 7)


def test_ray_color_depth_and_background(oracle):  # core.clj:19-30
    from raytracing_amd.core import build_camera
    from raytracing_amd.scenes import scene_config
    cfg = scene_config("cornell-box")
    cam = build_camera(cfg, 1.0)
    empty = flatten.flatten_world([])
    p = abi.make_params(8, 8, 2, 50, (0.7, 0.8, 1.0))
    rgb, argb, rays = oracle.render(empty, cam, p)
    assert np.allclose(rgb, np.array([1.4, 1.6, 2.0])) and rays == 128   # every path: one miss -> background
    p0 = abi.make_params(8, 8, 2, 0, (0.7, 0.8, 1.0))
    rgb, _, rays = oracle.render(empty, cam, p0)
    assert (rgb == 0).all() and rays == 0                                  # depth <= 0 -> black, no hit test
    # a light in front of the camera: emitted only, no attenuation chain
    lit = flatten.flatten_world([xy_rect(-1e4, 1e4, -1e4, 1e4, 100, diffuse_light([4, 5, 6]))])
    rgb, _, _ = oracle.render(lit, cam, abi.make_params(4, 4, 3, 50, (0, 0, 0)))
    assert np.allclose(rgb, np.array([12.0, 15.0, 18.0]))
    # depth 1: the first scatter succeeds, the recursive call returns black
    wall = flatten.flatten_world([xy_rect(-1e4, 1e4, -1e4, 1e4, 100, lambertian([1, 1, 1]))])
    rgb, _, rays = oracle.render(wall, cam, abi.make_params(4, 4, 3, 1, (1, 1, 1)))
    assert (rgb == 0).all() and rays == 48
