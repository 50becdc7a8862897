"""Progressive / checkpointed accumulation of a long render.

A sample's radiance depends only on (seed, pixel, global sample index), so a job of `spp` samples
per pixel can be rendered as consecutive slices of samples (`rt_params.sample_offset`), the FP32
sums added up, and the pair (sums, next sample index) saved between slices: a 4K frame at 10 000 spp
becomes restartable and previewable. The reference has nothing of the kind (its `-main` renders in
one go, core.clj:96-127); the per-pixel post-process of core.clj:106-108 is applied to the sums at
any time with the number of samples done so far as the divisor.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np

from . import abi


def slice_params(base: abi.rt_params, first: int, count: int) -> abi.rt_params:
    """rt_params for samples [first, first + count) of the job `base` describes."""
    p = abi.rt_params()
    for name, _ in abi.rt_params._fields_:
        if name != "background":
            setattr(p, name, getattr(base, name))
    p.background[:] = list(base.background)
    p.sample_offset = base.sample_offset + first
    p.spp = count
    return p


def job_digest(flat, cam: abi.rt_camera, base: abi.rt_params, slice_spp: int) -> str:
    """What a checkpoint must agree on before its sums may be continued: the world (its scene-file
    bytes), the camera, the background and the slice size (the FP32 sums depend on how the samples
    were sliced). Width, height, spp, depth and seed are compared field by field besides."""
    import hashlib
    from . import scenefile
    h = hashlib.sha256()
    h.update(scenefile.dumps(flat))
    h.update(bytes(cam))
    h.update(np.asarray(list(base.background), dtype=np.float64).tobytes())
    h.update(np.int64(slice_spp).tobytes())
    return h.hexdigest()


def save_checkpoint(path: str, rgb_sum: np.ndarray, done: int, base: abi.rt_params, digest: str = "") -> None:
    tmp = path + ".tmp.npz"
    np.savez(tmp, rgb_sum=np.asarray(rgb_sum, dtype=np.float32), done=np.int64(done),
             job=np.array([base.width, base.height, base.spp, base.sample_offset, base.max_depth], dtype=np.int64),
             seed=np.uint64(base.seed), digest=np.array(digest))
    os.replace(tmp, path)  # a crash never leaves a torn checkpoint behind


def load_checkpoint(path: str, base: abi.rt_params, digest: str = ""):
    """-> (rgb_sum, done) of a checkpoint written for the same job, else raises ValueError."""
    with np.load(path, allow_pickle=False) as z:
        job = [int(v) for v in z["job"]]
        if job != [base.width, base.height, base.spp, base.sample_offset, base.max_depth] or int(z["seed"]) != int(base.seed):
            raise ValueError(f"checkpoint {path} belongs to another job: {job}")
        saved = str(z["digest"]) if "digest" in z.files else ""
        if saved != digest:
            raise ValueError(f"checkpoint {path} was written for another world, camera, background or slice size")
        return z["rgb_sum"].astype(np.float32), int(z["done"])


def render_progressive(render_fn: Callable[[abi.rt_params], np.ndarray], base: abi.rt_params, slice_spp: int,
                       checkpoint: Optional[str] = None,
                       on_slice: Optional[Callable[[np.ndarray, int], None]] = None, digest: str = "",
                       flat=None, cam: Optional[abi.rt_camera] = None):
    """Renders `base` in slices of `slice_spp` samples. render_fn(params) -> H x W x 3 FP32 sums of
    that slice. Resumes from `checkpoint` when it exists and matches the job; writes it after every
    slice. on_slice(rgb_sum, samples_done) is called after every slice (preview hook). A checkpoint
    is tied to the world, camera, background and slice size by a digest: pass the flattened world
    and the camera (`flat`, `cam`) and it is job_digest of them, or pass `digest` yourself. An
    existing checkpoint is never continued without one: its sums could be another world's.
    Returns (rgb_sum, samples_done)."""
    if slice_spp <= 0:
        raise ValueError("slice_spp must be positive")
    if not digest and flat is not None and cam is not None:
        digest = job_digest(flat, cam, base, slice_spp)
    acc, done = None, 0
    if checkpoint and os.path.exists(checkpoint):
        if not digest:
            raise ValueError(f"checkpoint {checkpoint} exists: continuing it needs flat= and cam= (or digest=), "
                             "so that it can be checked against this world, camera and slice size")
        acc, done = load_checkpoint(checkpoint, base, digest)
    while done < base.spp:
        n = min(slice_spp, base.spp - done)
        part = np.asarray(render_fn(slice_params(base, done, n)), dtype=np.float32)
        acc = part.copy() if acc is None else acc + part
        done += n
        if checkpoint:
            save_checkpoint(checkpoint, acc, done, base, digest)
        if on_slice:
            on_slice(acc, done)
    if acc is None:
        acc = np.zeros((base.height, base.width, 3), dtype=np.float32)
    return acc, done
