"""rt_render_device_noalloc (scratch in a fixed device array inside the library, no allocation)
against rt_render_device, through the C ABI: bit-identical sums and pixels on two calls in a row,
the second with a smaller input (the array keeps the first call's contents), then a third on
another world, a request beyond the array's size, and a run on a side stream; then the call captured into a HIP
graph (one of the callers it exists for) and replayed."""
import numpy as np
import pytest

from raytracing_amd import abi
from raytracing_amd.lib import RtError
from util import scene_bundle

pytestmark = pytest.mark.gpu


def _both(sc, cam, p, stream=None):
    import torch
    H, W = p.height, p.width
    out = []
    for noalloc in (False, True):
        rgb = torch.full((H, W, 3), -7.0, dtype=torch.float32, device="cuda:0")
        argb = torch.full((H, W), 0x55, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        s = stream.cuda_stream if stream is not None else 0
        if noalloc:
            sc.render_device_noalloc(cam, p, rgb.data_ptr(), argb.data_ptr(), stream=s)
        else:
            sc.render_device(cam, p, rgb.data_ptr(), argb.data_ptr(), stream=s)
        if stream is not None:
            stream.synchronize()
        torch.cuda.synchronize()
        out.append((rgb.cpu().numpy(), argb.cpu().numpy().view(np.uint32)))
    return out


def test_noalloc_matches_render_device(gpu_ctx):
    import torch
    cfg, flat, cam = scene_bundle("cornell-box")
    sc = gpu_ctx.scene(flat)
    # first call: 200x120 at 150 spp (tiered schedule: 375 tiles x 64 x ~112 chunks = 2.7 M work items)
    big = abi.make_params(200, 120, 150, 50, cfg["background"], seed=41)
    (a_rgb, a_argb), (b_rgb, b_argb) = _both(sc, cam, big)
    assert np.array_equal(a_rgb.view(np.uint32), b_rgb.view(np.uint32)) and np.array_equal(a_argb, b_argb)
    assert (a_rgb >= 0).all() and a_rgb.max() > 0
    # second call, smaller in every way (fewer pixels, fewer chunks, a shard, a sample offset): the
    # array still holds the first call's partial sums and queue heads
    small = abi.make_params(37, 19, 5, 50, cfg["background"], seed=42, sample_offset=3, shard_index=1, shard_count=2)
    (a_rgb, a_argb), (b_rgb, b_argb) = _both(sc, cam, small)
    assert np.array_equal(a_rgb.view(np.uint32), b_rgb.view(np.uint32)) and np.array_equal(a_argb, b_argb)
    assert a_rgb.max() > 0 and (a_rgb == 0).any()     # the other shard's pixels are written as 0
    # only one output asked for; and what the entry point is for: the device's free memory is the same before and
    # after a run of calls (hipMemGetInfo: an allocation by the library, however it were made, would show)
    rgb = torch.zeros((19, 37, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(4):
        sc.render_device_noalloc(cam, small, rgb.data_ptr(), 0)
    free_after, _ = torch.cuda.mem_get_info()
    torch.cuda.synchronize()
    assert free_after == free_before, f"free device memory moved by {free_before - free_after} bytes across noalloc calls"
    assert np.array_equal(rgb.cpu().numpy().view(np.uint32), a_rgb.view(np.uint32))
    # beyond the array: refused with a message that says why, nothing launched
    with pytest.raises(RtError) as e:
        sc.render_device_noalloc(cam, abi.make_params(1024, 1024, 1000, 50, cfg["background"]), rgb.data_ptr(), 0)
    assert e.value.code == abi.RT_ERR_UNSUPPORTED and "work items" in str(e.value)
    with pytest.raises(RtError):
        sc.render_device_noalloc(cam, abi.make_params(16, 16, 2, 50, cfg["background"], flags=abi.RT_FLAG_COUNTERS), rgb.data_ptr(), 0)
    sc.close()
    # another world (BVH, media, textures: another kernel instantiation, LDS stacks) on a side stream
    cfg, flat, cam = scene_bundle("final-scene")
    sc = gpu_ctx.scene(flat)
    p = abi.make_params(96, 64, 6, 50, cfg["background"], seed=43)
    (a_rgb, a_argb), (b_rgb, b_argb) = _both(sc, cam, p, stream=torch.cuda.Stream())
    assert np.array_equal(a_rgb.view(np.uint32), b_rgb.view(np.uint32)) and np.array_equal(a_argb, b_argb)
    sc.close()


def test_noalloc_call_is_capturable(gpu_ctx):
    """Stream capture forbids allocation and synchronisation: the call goes into a graph as it is (two memsets, the
    path kernel, resolve and tonemap) and every replay gives the frame of rt_render_device bit for bit."""
    import torch
    cfg, flat, cam = scene_bundle("cornell-box")
    sc = gpu_ctx.scene(flat)
    p = abi.make_params(64, 48, 12, 50, cfg["background"], seed=77)
    H, W = p.height, p.width
    want_rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    want_argb = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
    sc.render_device(cam, p, want_rgb.data_ptr(), want_argb.data_ptr())
    torch.cuda.synchronize()
    rgb = torch.full((H, W, 3), -7.0, dtype=torch.float32, device="cuda:0")
    argb = torch.full((H, W), 0x55, dtype=torch.int32, device="cuda:0")
    sc.render_device_noalloc(cam, p, rgb.data_ptr(), argb.data_ptr())      # first call on the device: the library's set-up is done
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sc.render_device_noalloc(cam, p, rgb.data_ptr(), argb.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        rgb.fill_(-7.0)
        argb.fill_(0x55)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(rgb.view(torch.int32), want_rgb.view(torch.int32)) and torch.equal(argb, want_argb)
    sc.close()
