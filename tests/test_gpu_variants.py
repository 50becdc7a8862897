"""Every kernel instantiation that covers a world answers a closest-hit query with the SAME
record, bit for bit.

rt_hit_batch runs the most general instantiation; the render of a world runs the smallest one
that covers it, which takes other template branches (record tables in LDS, no FP64 ray residuals
without chains, rect pair records fetched one or two at a time, media answered from one
intersection, ...). The 1e-5 hit-record parity is established against the oracle once
(tests/test_gpu_parity.py); this test carries it to every instantiation, the render's included.
"""
import os

import numpy as np
import pytest

from raytracing_amd import abi
from util import ALL_SCENES, medium_uniform, probe_rays, scene_bundle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEED = 0x5EED0001

# what each world renders on (instantiation index as in include/rtmi.h)
RENDERS_ON = {"cornell-box": 0, "cornell-smoke": 1, "two-spheres": 2, "earth": 2, "random-scene": 7,
              "two-perlin-spheres": 4, "simple-light": 4, "final-scene": 5}


@pytest.mark.parametrize("name", ALL_SCENES)
def test_records_are_bit_identical_across_instantiations(gpu_ctx, oracle, name):
    cfg, flat, cam = scene_bundle(name)
    sc = gpu_ctx.scene(flat)
    rv, cover = sc.variants()
    assert rv == RENDERS_ON[name] and rv in cover and abi.VARIANT_NAMES.index("V_FULL") in cover and len(cover) >= 2
    nm = flat.desc.n_media
    # probe rays, and rays that real paths produce (they start ON surfaces)
    n = 20000
    o, d, tm = probe_rays(cfg, n, seed=31)
    p = abi.make_params(64, 64, 4, 50, cfg["background"], seed=SEED)
    rng = np.random.default_rng(6)
    rays, xis = [], []
    for _ in range(400):
        px, smp = int(rng.integers(0, 64 * 64)), int(rng.integers(0, 4000))
        r = oracle.path_rays(flat, cam, p, px, smp)
        rays.append(r)
        for b in range(len(r)):
            xis.append([medium_uniform(SEED, px, smp, b, m) for m in range(nm)])
    R = np.concatenate(rays)
    o = np.concatenate([o, R[:, 0:3].astype(np.float32)])
    d = np.concatenate([d, R[:, 3:6].astype(np.float32)])
    tm = np.concatenate([tm, R[:, 6].astype(np.float32)])
    xi = None
    if nm:
        xi = np.concatenate([np.random.default_rng(3).random((n, nm)).astype(np.float32),
                             np.array(xis, dtype=np.float32).reshape(-1, nm)])
    for with_media in ((True, False) if nm else (False,)):
        x = xi if with_media else None
        base = sc.hit_batch(o, d, tm, medium_xi=x)          # rt_hit_batch
        assert int(base["hit"].sum()) > len(o) // 20
        for v in cover:
            got = sc.hit_batch(o, d, tm, medium_xi=x, variant=v)
            same = got.tobytes() == base.tobytes()
            if not same:
                bad = np.nonzero([a.tobytes() != b.tobytes() for a, b in zip(got, base)])[0]
                print(f"{name}: instantiation {abi.VARIANT_NAMES[v]} differs on {len(bad)} of {len(o)} rays, first:\n"
                      f"  {got[bad[0]]}\n  {base[bad[0]]}")
            assert same, f"{name}: {abi.VARIANT_NAMES[v]} differs from the general instantiation"
    # an instantiation that does not cover the world is refused, not run
    from raytracing_amd.lib import RtError
    for v in range(abi.RT_N_VARIANTS):
        if v not in cover:
            with pytest.raises(RtError) as e:
                sc.hit_batch(o[:4], d[:4], tm[:4], variant=v)
            assert e.value.code == abi.RT_ERR_UNSUPPORTED
    sc.close()


@pytest.mark.parametrize("name", ALL_SCENES)
def test_golden_records_on_the_render_instantiation(gpu_ctx, name):
    """The frozen oracle answers of tests/golden/ against the instantiation the render uses (no
    oracle at run time)."""
    from test_gpu_parity import _check_hits, _uv_exempt
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg, flat, cam = scene_bundle(name)
    sc = gpu_ctx.scene(flat)
    xi = g["xi"] if g["xi"].size else None
    dev = sc.hit_batch(g["orig"], g["dir"], g["time"], medium_xi=xi, variant=abi.RT_VARIANT_RENDER)
    _check_hits(dev, g["hits"], g["orig"], g["dir"], 0.0, _uv_exempt(name, flat))
    sc.close()
