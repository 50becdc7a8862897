"""oracle.cpp held against a second restatement of the reference written independently of it (oracle/second.py: pure
Python, the Clojure text form by form, over the host mirror's records): closest-hit records on all eight reference
worlds (media with scripted uniforms), every material's scatter and emission with scripted draws, every texture, the
three samplers. Both are restatements — the reference ships no tests and cannot run here — so this does not pin the
oracle to the reference; it shows that two readings of the same Clojure agree to rounding. No GPU."""
import random

import numpy as np
import pytest

from oracle import binding as ob
from oracle import second as S
from raytracing_amd import flatten, scenes
from raytracing_amd import hittables as H
from util import ALL_SCENES, probe_rays

REL = 1e-11


def _world(name, seed=2):
    cfg = scenes.scene_config(name)
    world = cfg["world"](random.Random(seed))
    f = flatten._Flattener()
    f.fs.root = f._list(None, list(world))
    flat = f.fs.finish()
    return cfg, world, f, flat


def _media(world):
    """ConstantMedium objects reachable from the world."""
    out, seen, stack = [], set(), list(world)
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if isinstance(o, H.ConstantMedium):
            out.append(o)
            stack.append(o.boundary)
        elif isinstance(o, (H.HittableList,)):
            stack.extend(o.objects)
        elif isinstance(o, H.Box):
            stack.extend(o.sides)
        elif isinstance(o, H.BvhNode):
            stack.extend([o.left, o.right])
        elif isinstance(o, (H.RotateY, H.Translate)):
            stack.append(o.ptr)
    return out


def _close(a, b, scale=1.0):
    return abs(a - b) <= REL * max(abs(a), abs(b), scale)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_closest_hit_records_agree_on_every_reference_world(name):
    cfg, world, f, flat = _world(name)
    n = 60 if name in ("final-scene", "random-scene") else 160
    o32, d32, tm32 = probe_rays(cfg, n, seed=4242)
    media = _media(world)
    index_of = {id(m): f.fs.node_i[f.node_of[id(m)]][2] for m in media}
    assert sorted(index_of.values()) == list(range(len(media))) and len(media) == flat.desc.n_media
    rng = np.random.default_rng(7)
    xi32 = rng.uniform(0.02, 0.98, (n, max(len(media), 1))).astype(np.float32)
    orc = ob.hit_batch(flat, o32, d32, tm32, medium_xi=xi32 if media else None)
    hits = misses = 0
    for i in range(n):
        r = S.Ray([float(v) for v in o32[i]], [float(v) for v in d32[i]], float(tm32[i]))
        rec = S._list_hit(world, r, 0.001, S.INF, lambda m, i=i: float(xi32[i, index_of[id(m)]]))
        row = orc[i]
        if rec is None:
            assert row[11] == 0.0, f"{name} ray {i}: oracle.cpp hits at t={row[0]}, the second restatement misses"
            misses += 1
            continue
        hits += 1
        assert row[11] == 1.0, f"{name} ray {i}: the second restatement hits at t={rec['t']}, oracle.cpp misses"
        size = max(1.0, float(np.abs(o32[i]).max()), abs(rec["t"]) * float(np.linalg.norm(d32[i])))
        assert _close(row[0], rec["t"], 1e-300), (name, i, row[0], rec["t"])
        for q in range(3):
            assert _close(row[1 + q], rec["p"][q], size), (name, i, "p", row[1:4], rec["p"])
            assert _close(row[4 + q], rec["normal"][q]), (name, i, "normal", row[4:7], rec["normal"])
        if rec["u"] is not None:
            for got, want in ((row[7], rec["u"]), (row[8], rec["v"])):
                assert (got != got and want != want) or _close(got, want), (name, i, "uv", row[7:9], rec["u"], rec["v"])
        assert bool(row[9]) == rec["front_face"]
        assert int(row[10]) == f.mat_of[id(rec["mat_ptr"])]
    assert hits >= 10, (name, hits, misses)


def _script(rng, n=96):
    return [float(x) for x in rng.random(n)]


class _Scripted:
    def __init__(self, script):
        self.script, self.pos = script, 0

    def __call__(self, *_):
        v = self.script[self.pos]
        self.pos += 1
        return v


@pytest.mark.parametrize("name", ["random-scene", "cornell-smoke", "final-scene", "two-perlin-spheres", "earth", "simple-light"])
def test_scatter_and_emission_agree_for_every_material_met(name):
    """A record per probe ray that hits (the second restatement's own), then (scatter mat r rec) and (emitted mat u v p)
    on both sides with the same scripted uniforms: ok flag, attenuation, scattered ray, emission, and the NUMBER of
    uniforms consumed (the rejection loops and the dielectric's short-circuit draw)."""
    cfg, world, f, flat = _world(name)
    o32, d32, tm32 = probe_rays(cfg, 120, seed=99)
    rng = np.random.default_rng(3)
    seen_kinds = set()
    for i in range(len(o32)):
        r = S.Ray([float(v) for v in o32[i]], [float(v) for v in d32[i]], float(tm32[i]))
        rec = S._list_hit(world, r, 0.001, S.INF, lambda m: 0.37)
        if rec is None:
            continue
        mat = rec["mat_ptr"]
        seen_kinds.add(type(mat).__name__)
        script = _script(rng)
        draws = _Scripted(script)
        s = S.scatter(mat, r, rec, draws)
        em = S.emitted(mat, rec["u"], rec["v"], rec["p"])
        u, v = (0.0, 0.0) if rec["u"] is None else (rec["u"], rec["v"])
        if u != u or v != v:
            continue   # (a NaN u/v indexes texel 0 on both sides; the scripted hook takes plain numbers)
        got = ob.scatter_scripted(flat, f.mat_of[id(mat)], list(r.orig) + list(r.dir) + [r.time_],
                                  list(rec["p"]) + list(rec["normal"]) + [u, v, 1.0 if rec["front_face"] else 0.0], script)
        assert got["used"] == draws.pos, (name, i, type(mat).__name__, got["used"], draws.pos)
        assert got["ok"] == s["ok"]
        for q in range(3):
            assert _close(got["attenuation"][q], s["attenuation"][q]), (name, i, "attenuation")
            assert _close(got["emitted"][q], em[q]), (name, i, "emitted")
            assert _close(got["orig"][q], s["scattered"].orig[q], max(1.0, abs(s["scattered"].orig[q]))), (name, i, "orig")
            assert _close(got["dir"][q], s["scattered"].dir[q]), (name, i, "dir", got["dir"], s["scattered"].dir)
        assert got["time"] == s["scattered"].time_
    assert seen_kinds, name


def test_every_material_kind_was_met_somewhere():
    kinds = set()
    for name in ("random-scene", "cornell-smoke", "simple-light", "final-scene"):
        cfg, world, f, flat = _world(name)
        o32, d32, tm32 = probe_rays(cfg, 120, seed=99)
        for i in range(len(o32)):
            r = S.Ray([float(v) for v in o32[i]], [float(v) for v in d32[i]], float(tm32[i]))
            rec = S._list_hit(world, r, 0.001, S.INF, lambda m: 0.37)
            if rec is not None:
                kinds.add(type(rec["mat_ptr"]).__name__)
    assert kinds == {"Lambertian", "Metal", "Dielectric", "DiffuseLight", "Isotropic"}, kinds


@pytest.mark.parametrize("name", ["random-scene", "two-perlin-spheres", "earth", "final-scene"])
def test_texture_values_agree(name):
    """Solid colours, the checker (random-scene's ground), Perlin marble at both scales the reference uses (4 and the
    0.1 that `^long` makes 0) and the image texture, at random (u, v, p) and at the clamped edges."""
    cfg, world, f, flat = _world(name)
    rng = np.random.default_rng(11)
    textures = {idx: obj for obj in f._alive for idx in [f.tex_of.get(id(obj))] if idx is not None}
    assert textures
    kinds = set()
    for idx, tex in textures.items():
        kinds.add(type(tex).__name__)
        for trial in range(12):
            u, v = (float(x) for x in rng.uniform(-0.1, 1.1, 2))
            if trial == 0:
                u, v = 0.0, 1.0
            if trial == 1:
                u, v = 1.0, 0.0
            p = [float(x) for x in rng.uniform(-300, 300, 3)] if trial % 2 else [float(x) for x in rng.uniform(-3, 3, 3)]
            got = ob.texture_value(flat, idx, u, v, p)
            want = S.value(tex, u, v, p)
            for q in range(3):
                assert abs(got[q] - want[q]) <= 1e-9, (name, type(tex).__name__, u, v, p, got, want)
    assert kinds & {"NoiseTexture", "ImageTexture", "CheckerTexture", "SolidColor"}


def test_perlin_noise_and_turbulence_agree():
    cfg, world, f, flat = _world("two-perlin-spheres")
    perlins = {idx: obj for obj in f._alive for idx in [f.perlin_of.get(id(obj))] if idx is not None}
    assert perlins
    rng = np.random.default_rng(5)
    for pid, per in perlins.items():
        for _ in range(40):
            p = [float(x) for x in rng.uniform(-40, 40, 3)]
            assert abs(ob.perlin_noise(flat, pid, p) - S.noise(per, p)) <= 1e-12
            assert abs(ob.perlin_turb(flat, pid, p, 7) - S.turb(per, p, 7)) <= 1e-12


@pytest.mark.parametrize("which,fn", [(0, S.random_in_unit_sphere), (1, S.random_in_unit_disk), (2, S.random_unit_vector)])
def test_samplers_agree_draw_for_draw(which, fn):
    rng = np.random.default_rng(17 + which)
    for _ in range(200):
        script = _script(rng, 120)
        draws = _Scripted(script)
        want = fn(draws)
        got, used = ob.sampler_scripted(which, script)
        assert used == draws.pos
        for q in range(3):
            assert _close(got[q], want[q])


@pytest.mark.parametrize("name", ALL_SCENES)
def test_whole_samples_agree_with_scripted_draws(name):
    """core.clj:96-105 end to end — the two jitters, get-ray (lens disk, shutter time), ray-color to its depth limit
    with media, scatter and textures on the way — on both sides with ONE scripted sequence of uniforms consumed in
    the reference's evaluation order: the radiance of the sample and the number of uniforms consumed must agree, and
    the camera map of camera.clj must be the one the host mirror hands the library."""
    from raytracing_amd import abi
    from raytracing_amd.core import build_camera
    cfg, world, f, flat = _world(name)
    W = H = 40
    cam_abi = build_camera(cfg, 1.0)
    cam = S.camera(cfg["lookfrom"], cfg["lookat"], (0.0, 1.0, 0.0), cfg["vfov"], 1.0, cfg["aperture"], 10.0, 0.0, 1.0)
    for key, field in (("origin", cam_abi.origin), ("lower_left_corner", cam_abi.lower_left_corner), ("horizontal", cam_abi.horizontal),
                       ("vertical", cam_abi.vertical), ("u", cam_abi.u), ("v", cam_abi.v), ("w", cam_abi.w)):
        for q in range(3):
            assert _close(cam[key][q], field[q], 1.0), (name, key)
    assert cam["lens_radius"] == cam_abi.lens_radius and (cam["time0"], cam["time1"]) == (cam_abi.time0, cam_abi.time1)
    params = abi.make_params(W, H, 1, 50, cfg["background"], seed=1)
    rng = np.random.default_rng(23)
    n = 12 if name in ("final-scene", "random-scene") else 40
    lit = 0
    for _ in range(n):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        script = [float(v) for v in rng.random(4000)]
        draws = _Scripted(script)
        want = S.sample(cam, W, H, x, y, cfg["background"], world, 50, draws)
        got, used = ob.sample_scripted(flat, cam_abi, params, x, y, script)
        assert used == draws.pos, (name, x, y, used, draws.pos)
        for q in range(3):
            assert _close(got[q], want[q], 1e-6), (name, x, y, got, want)
        lit += any(c > 0 for c in want)
    assert lit > 0, name


def test_pixel_packing_agrees():
    rng = np.random.default_rng(31)
    cases = [(0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.25, 1.5, float("nan")), (1e-9, 0.998001, 0.998002), (4.0, 0.5, 0.0625), (-0.0, 0.0, 3e38)]
    cases += [tuple(float(v) for v in rng.uniform(0, 1.3, 3)) for _ in range(300)]
    for spp in (1, 7, 1000):
        for c in cases:
            acc = [v * spp for v in c]
            assert ob.pack_pixel(acc[0], acc[1], acc[2], spp) == S.pack_pixel(acc, spp), (c, spp)
    assert S.pack_pixel([0.25, 1.5, float("nan")], 1) == 0xFF80FF00      # the known answer of the survey's K13
