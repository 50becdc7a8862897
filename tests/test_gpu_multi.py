"""rt_multi_*: several GPUs behind one handle for a single-process caller (the stand-in for the
reference's thread-per-row fan-out and latch, core.clj:109-128).

A box with ONE device runs the one-device path and the fan-out rehearsed with one device named several
times (shards, ordering, the sum on the first device, the post-process). The RCCL reduce between
DISTINCT devices runs in test_distinct_devices_reduce_over_rccl wherever two or more devices are visible
and is skipped, saying so, where there is one."""
import os

import numpy as np
import pytest

from raytracing_amd import abi, lib
from util import scene_bundle

pytestmark = pytest.mark.gpu


def test_one_device_equals_rt_render(gpu_ctx):
    cfg, flat, cam = scene_bundle("cornell-box")
    p = abi.make_params(70, 50, 9, 50, cfg["background"], seed=12)
    sc = gpu_ctx.scene(flat)
    rgb, argb, st = sc.render(cam, p)
    sc.close()
    m = lib.MultiContext([0])
    assert m.device_count() == 1
    ms = m.scene(flat)
    for split in (abi.RT_SPLIT_TILES, abi.RT_SPLIT_SAMPLES):
        r2, a2, s2 = ms.render(cam, p, split)
        assert np.array_equal(r2, rgb) and np.array_equal(a2, argb) and s2["rays"] == st["rays"] and s2["samples"] == st["samples"]
    ms.close()
    m.close()


@pytest.mark.parametrize("name,n", [("cornell-box", 2), ("cornell-box", 3), ("final-scene", 4), ("two-perlin-spheres", 8)])
def test_fan_out_over_a_repeated_device(gpu_ctx, name, n):
    cfg, flat, cam = scene_bundle(name)
    W, H, spp = 61, 43, 13
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=31)
    sc = gpu_ctx.scene(flat)
    rgb, argb, st = sc.render(cam, p)
    sc.close()
    m = lib.MultiContext([0] * n)
    ms = m.scene(flat)
    # tiles: disjoint pixels, every pixel summed in the single-GPU order -> the same bits
    r2, a2, s2 = ms.render(cam, p, abi.RT_SPLIT_TILES)
    assert np.array_equal(r2, rgb) and np.array_equal(a2, argb)
    assert s2["rays"] == st["rays"] and s2["samples"] == W * H * spp
    # sample slices: the same samples, FP32 sums re-associated
    r3, a3, s3 = ms.render(cam, p, abi.RT_SPLIT_SAMPLES)
    np.testing.assert_allclose(r3, rgb, rtol=1e-5, atol=1e-5)
    assert s3["rays"] == st["rays"] and s3["samples"] == W * H * spp
    ch = np.array([16, 8, 0])[:, None, None]
    assert np.abs(((a3 >> ch) & 255).astype(int) - ((argb >> ch) & 255).astype(int)).max() <= 1
    # fewer samples than devices: the devices left without a slice contribute zeros
    few = abi.make_params(W, H, 2, 50, cfg["background"], seed=31)
    sc = gpu_ctx.scene(flat)
    ref, _, _ = sc.render(cam, few)
    sc.close()
    r4, _, s4 = ms.render(cam, few, abi.RT_SPLIT_SAMPLES)
    np.testing.assert_allclose(r4, ref, rtol=1e-5, atol=1e-5)
    assert s4["samples"] == W * H * 2
    ms.close()
    m.close()


def _n_devices():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("name", ["cornell-box", "final-scene"])
def test_distinct_devices_reduce_over_rccl(gpu_ctx, name):
    """rt_multi_render over distinct device ids: one host thread per device, the frames summed onto the first
    device by one grouped ncclReduce (core.clj:109-128 fans rows out to threads and waits on a latch). The tile
    split must give the single-GPU frame bit for bit (disjoint pixels, zeros elsewhere), the sample split the
    same samples with the FP32 sums re-associated."""
    n = _n_devices()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the reduce between distinct devices needs two or more "
                    "(the repeated-device tests above cover everything but RCCL itself)")
    ids = list(range(min(n, 8)))
    cfg, flat, cam = scene_bundle(name)
    W, H, spp = 131, 77, 2 * len(ids) + 1
    p = abi.make_params(W, H, spp, 50, cfg["background"], seed=77)
    sc = gpu_ctx.scene(flat)
    rgb, argb, st = sc.render(cam, p)
    sc.close()
    for use in ([0, 1], ids):
        m = lib.MultiContext(use)
        assert m.device_count() == len(use)
        ms = m.scene(flat)
        r2, a2, s2 = ms.render(cam, p, abi.RT_SPLIT_TILES)
        assert np.array_equal(r2, rgb) and np.array_equal(a2, argb)
        assert s2["rays"] == st["rays"] and s2["samples"] == W * H * spp
        r3, a3, s3 = ms.render(cam, p, abi.RT_SPLIT_SAMPLES)
        np.testing.assert_allclose(r3, rgb, rtol=1e-5, atol=1e-5)
        assert s3["rays"] == st["rays"] and s3["samples"] == W * H * spp
        # twice through the same communicator: the second reduce must not see the first one's frame
        r4, a4, _ = ms.render(cam, p, abi.RT_SPLIT_TILES)
        assert np.array_equal(r4, rgb) and np.array_equal(a4, argb)
        ms.close()
        m.close()


def test_multi_errors(gpu_ctx):
    with pytest.raises(lib.RtError):
        lib.MultiContext([])
    with pytest.raises(lib.RtError):
        lib.MultiContext([0, 99])
    cfg, flat, cam = scene_bundle("cornell-box")
    m = lib.MultiContext([0, 0])
    ms = m.scene(flat)
    with pytest.raises(lib.RtError) as e:
        ms.render(cam, abi.make_params(16, 16, 2, 50, cfg["background"], shard_index=1, shard_count=2))
    assert e.value.code == abi.RT_ERR_ARG
    with pytest.raises(lib.RtError):
        ms.render(cam, abi.make_params(16, 16, 2, 50, cfg["background"]), split=7)
    ms.close()
    m.close()


def test_rccl_loads_and_reduces_on_one_rank(gpu_ctx):
    """librccl.so resolves, and the calls the multi-device reduce makes run on one rank."""
    L = lib.load()
    rc = L.rt_multi_selftest_rccl(0)
    assert rc == 0, L.rt_multi_last_error().decode()


def test_main_over_several_devices_gives_the_frame_of_one(tmp_path):
    """raytracing_amd.core.main with `devices` (the :gpus key the front door gains): a tile split over one device named
    three times returns the single-device frame bit for bit, a sample split the same picture to within one 8-bit step,
    and the command line writes the same frame as a PPM."""
    import subprocess
    import sys
    from raytracing_amd import core
    one, st1 = core.main(samples=6, scene="cornell-box", image_width=72)
    tiles, st3 = core.main(samples=6, scene="cornell-box", image_width=72, devices=[0, 0, 0], split="tile")
    assert np.array_equal(one, tiles) and st3["samples"] == st1["samples"] == 72 * 72 * 6
    slices, _ = core.main(samples=6, scene="cornell-box", image_width=72, devices=[0, 0], split="sample")
    ch = np.array([16, 8, 0])[:, None, None]
    assert np.abs(((one >> ch) & 255).astype(int) - ((slices >> ch) & 255).astype(int)).max() <= 1
    with pytest.raises(ValueError):
        core.main(samples=1, devices=[0, 0], split="rows")
    out = tmp_path / "frame.ppm"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, "-m", "raytracing_amd.core", "--samples", "6", "--scene", "cornell-box", "--width", "72",
                        "--gpus", "0,0,0", "--output", str(out)], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-600:]
    data = out.read_bytes()
    assert data.startswith(b"P6\n72 72\n255\n")
    px = np.frombuffer(data[len(b"P6\n72 72\n255\n"):], np.uint8).reshape(72, 72, 3)[::-1]
    assert np.array_equal(px[:, :, 0], (one >> 16) & 255) and np.array_equal(px[:, :, 2], one & 255)
