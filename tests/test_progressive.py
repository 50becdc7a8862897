"""Progressive / checkpointed accumulation (raytracing_amd/progressive.py), with the CPU oracle as the
renderer: slices add up to the one-shot render, and a job resumed from its checkpoint ends with
exactly the sums of the uninterrupted one."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_amd import abi  # noqa: E402
from raytracing_amd.core import build_camera, setup  # noqa: E402
from raytracing_amd.progressive import load_checkpoint, render_progressive, slice_params  # noqa: E402

W, H, SPP = 24, 16, 10


@pytest.fixture(scope="module")
def job(oracle):
    cfg, flat = setup("cornell-box", scene_seed=1)
    cam = build_camera(cfg, W / H)
    base = abi.make_params(W, H, SPP, 50, cfg["background"], seed=77)

    def render_fn(p):
        rgb, _, _ = oracle.render(flat, cam, p, nthreads=2, want_argb=False)
        return np.asarray(rgb, dtype=np.float32).reshape(H, W, 3)
    return base, render_fn


def test_slices_add_up_to_the_one_shot_render(job):
    base, render_fn = job
    whole = render_fn(base)
    seen = []
    acc, done = render_progressive(render_fn, base, 3, on_slice=lambda a, d: seen.append(d))
    assert done == SPP and seen == [3, 6, 9, 10]
    np.testing.assert_allclose(acc, whole, rtol=2e-6, atol=1e-6)


def test_resume_from_checkpoint_is_exact(job, tmp_path):
    base, render_fn = job
    straight, _ = render_progressive(render_fn, base, 4)
    ck = str(tmp_path / "job.npz")
    calls = []

    def interrupted(p):
        if len(calls) == 2:
            raise KeyboardInterrupt
        calls.append((p.sample_offset, p.spp))
        return render_fn(p)
    with pytest.raises(KeyboardInterrupt):
        render_progressive(interrupted, base, 4, checkpoint=ck, digest="job-a")
    _, done = load_checkpoint(ck, base, "job-a")
    assert done == 8 and calls == [(0, 4), (4, 4)]
    with pytest.raises(ValueError):   # an existing checkpoint is not continued blind
        render_progressive(render_fn, base, 4, checkpoint=ck)
    with pytest.raises(ValueError):   # nor for a job that calls itself something else
        render_progressive(render_fn, base, 4, checkpoint=ck, digest="job-b")
    resumed, done = render_progressive(render_fn, base, 4, checkpoint=ck, digest="job-a")
    assert done == SPP
    np.testing.assert_array_equal(resumed, straight)


def test_checkpoint_of_another_job_is_refused(job, tmp_path):
    base, render_fn = job
    ck = str(tmp_path / "job.npz")
    render_progressive(render_fn, base, SPP, checkpoint=ck)
    other = abi.make_params(W, H, SPP, 50, (0.0, 0.0, 0.0), seed=78)
    with pytest.raises(ValueError):
        load_checkpoint(ck, other)
    p = slice_params(base, 4, 3)
    assert (p.sample_offset, p.spp, p.width, p.seed) == (4, 3, W, 77)
    with pytest.raises(ValueError):
        render_progressive(render_fn, base, 0)
