"""Progressive / checkpointed accumulation driven through the HIP path (Scene.render as the
renderer): slices add up to the one-shot frame, a job resumed from its checkpoint ends with exactly
the sums of the uninterrupted one, and a checkpoint of another world is refused."""
import numpy as np
import pytest

from raytracing_amd import abi
from raytracing_amd.progressive import job_digest, load_checkpoint, render_progressive
from util import scene_bundle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cornell-box", "final-scene"])
def test_progressive_on_the_gpu(gpu_ctx, name, tmp_path):
    cfg, flat, cam = scene_bundle(name)
    sc = gpu_ctx.scene(flat)
    W, H, SPP, SL = 72, 40, 23, 6
    base = abi.make_params(W, H, SPP, 50, cfg["background"], seed=91)
    calls = []

    def render_fn(p):
        calls.append((p.sample_offset, p.spp))
        return sc.render(cam, p, want_argb=False)[0]

    whole = render_fn(base)
    calls.clear()
    seen = []
    acc, done = render_progressive(render_fn, base, SL, on_slice=lambda a, d: seen.append(d))
    assert done == SPP and seen == [6, 12, 18, 23] and calls == [(0, 6), (6, 6), (12, 6), (18, 5)]
    np.testing.assert_allclose(acc, whole, rtol=1e-5, atol=1e-5)      # same samples, FP32 sums re-associated
    # interrupted after two slices, resumed from the checkpoint: bit-identical to the straight run
    dg = job_digest(flat, cam, base, SL)
    ck = str(tmp_path / "job.npz")
    n = [0]

    def interrupted(p):
        if n[0] == 2:
            raise KeyboardInterrupt
        n[0] += 1
        return render_fn(p)
    with pytest.raises(KeyboardInterrupt):
        render_progressive(interrupted, base, SL, checkpoint=ck, digest=dg)
    part, d = load_checkpoint(ck, base, dg)
    assert d == 12
    resumed, done = render_progressive(render_fn, base, SL, checkpoint=ck, flat=flat, cam=cam)   # the digest formed inside
    assert done == SPP
    np.testing.assert_array_equal(resumed, acc)
    # the checkpoint of this job is refused for another world, another camera, another slice size
    cfg2, flat2, cam2 = scene_bundle("two-spheres")
    for other in (job_digest(flat2, cam, base, SL), job_digest(flat, cam2, base, SL), job_digest(flat, cam, base, SL + 1)):
        assert other != dg
        with pytest.raises(ValueError):
            load_checkpoint(ck, base, other)
    # the pixel post-process applies to a partial job with the samples done so far as the divisor
    import torch
    t = torch.from_numpy(np.ascontiguousarray(part)).to("cuda:0")
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda:0")
    gpu_ctx.tonemap_device(t.data_ptr(), W * H, d, out.data_ptr())
    torch.cuda.synchronize()
    assert (out.cpu().numpy().view(np.uint32) >> 24 == 0xFF).all()
    sc.close()
