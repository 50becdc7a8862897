"""The multi-GPU driver on CPU: world_size 2 over the `gloo` backend.

`raytracing_amd.dist` only splits the job and issues one reduce; the renderer behind it is a
callable. Here that callable is the CPU oracle (test infrastructure), so what is under test is
the host logic a GPU run relies on: the shard arithmetic of `shard_params`, the zero-filled
full frames, the single reduce(sum) to rank 0, and that the split never changes a pixel
(the random stream of a sample depends only on (seed, pixel, global sample index)).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_amd import abi  # noqa: E402
from raytracing_amd.core import build_camera, setup  # noqa: E402
from raytracing_amd.dist import render_distributed, shard_params  # noqa: E402

W, H, SPP, SEED = 40, 24, 6, 0x5EED0001


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_render(flat, cam, p):
    from oracle import binding as ob
    rgb, _, _ = ob.render(flat, cam, p, nthreads=1, want_argb=False)
    return torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.float32).reshape(H, W, 3))


def _worker(rank, world, port, mode, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, flat = setup("cornell-box", scene_seed=1)
        cam = build_camera(cfg, W / H)
        base = abi.make_params(W, H, SPP, 50, cfg["background"], seed=SEED)
        acc = render_distributed(lambda p: _oracle_render(flat, cam, p), base, rank, world, mode)
        if rank == 0:
            np.save(out_path, acc.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _single():
    cfg, flat = setup("cornell-box", scene_seed=1)
    cam = build_camera(cfg, W / H)
    base = abi.make_params(W, H, SPP, 50, cfg["background"], seed=SEED)
    return _oracle_render(flat, cam, base).numpy()


@pytest.mark.parametrize("mode", ["tile", "sample"])
def test_two_ranks_gloo_match_single_process(tmp_path, mode):
    out = str(tmp_path / f"acc_{mode}.npy")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    got = np.load(out)
    want = _single()
    assert got.shape == want.shape
    if mode == "tile":
        # disjoint pixels + zeros: the reduce is a gather, so every pixel is bit-identical
        assert np.array_equal(got, want)
    else:
        # same samples, summed as (first half) + (second half) instead of one running sum
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    assert float(want.sum()) > 0.0


def test_shard_params_cover_job_exactly_once():
    base = abi.make_params(64, 48, 10, 50, (0.0, 0.0, 0.0), seed=7)
    for world in (1, 2, 3, 8):
        tiles = [shard_params(base, r, world, "tile") for r in range(world)]
        assert [(p.shard_index, p.shard_count, p.spp, p.sample_offset) for p in tiles] == \
               [(r, world, 10, 0) for r in range(world)]
        slices = [shard_params(base, r, world, "sample") for r in range(world)]
        covered = []
        for p in slices:
            assert (p.shard_index, p.shard_count) == (0, 1)
            covered += list(range(p.sample_offset, p.sample_offset + p.spp))
        assert covered == list(range(10))
    with pytest.raises(ValueError):
        shard_params(base, 0, 2, "rows")


def test_more_ranks_than_samples_contributes_zeros():
    base = abi.make_params(8, 8, 1, 50, (0.0, 0.0, 0.0), seed=7)
    calls = []

    def render_fn(p):
        calls.append(p.spp)
        return torch.ones(8, 8, 3)

    acc = render_distributed(render_fn, base, rank=0, world=2, mode="sample", reduce_fn=lambda t: None)
    assert calls == [1] and float(acc.abs().sum()) == 0.0


def tile_shard(x, y, width, count):
    """Which shard renders pixel (x, y): the rule of rt_params in include/rtmi.h."""
    tiles_x = (width + 7) // 8
    t = (y // 8) * tiles_x + x // 8
    turn = ((t // count) * count // tiles_x) % count
    return (t % count + count - turn) % count


@pytest.mark.parametrize("w,h", [(40, 24), (9, 33), (130, 17), (1, 1)])
def test_tile_shards_partition_the_frame_by_the_documented_rule(w, h):
    """The CPU restatement's tile shards: each renders exactly the pixels the rule gives it (zeros elsewhere), their
    sum is the whole frame bit for bit, and over a frame of many tile rows no shard keeps a fixed tile column."""
    from oracle import binding as ob
    cfg, flat = setup("cornell-box", scene_seed=1)
    cam = build_camera(cfg, w / h)

    def frame(**kw):
        p = abi.make_params(w, h, 2, 50, cfg["background"], seed=SEED, **kw)
        rgb, _, _ = ob.render(flat, cam, p, nthreads=1, want_argb=False)
        return np.ascontiguousarray(rgb, dtype=np.float32).reshape(h, w, 3)

    full = frame()
    ys, xs = np.mgrid[0:h, 0:w]
    for count in (2, 3, 8):
        owner = np.vectorize(lambda x, y: tile_shard(x, y, w, count))(xs, ys)
        acc = np.zeros_like(full)
        for idx in range(count):
            part = frame(shard_index=idx, shard_count=count)
            assert (part[owner != idx] == 0).all()
            assert np.array_equal(part[owner == idx], full[owner == idx])
            acc += part
        assert np.array_equal(acc, full)
    # 128 tiles per row, eight shards: the column a shard owns moves on from one tile row to the next
    cols = {tuple(x for x in range(0, 64, 8) if tile_shard(x, y, 1024, 8) == 0) for y in range(0, 64, 8)}
    assert len(cols) == 8
