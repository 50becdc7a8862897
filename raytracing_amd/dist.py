"""Multi-GPU driver: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The sample loop shards naturally: every (pixel, sample) is independent and its random stream is
keyed by (pixel, sample), so any split reproduces the same per-sample radiance. Two splits:
  * "tile":   rank r renders one 8x8 tile of every group of `world` consecutive tiles (all samples;
              the place within the group moves on by one per tile row, see rt_params in rtmi.h);
              pixels are disjoint, so summing the zero-filled full frames is a gather.
  * "sample": rank r renders a contiguous slice of the samples for every pixel.
Either way the only exchange is one reduce(sum) of the W*H*3 FP32 accumulators to rank 0,
which then applies the pixel post-process (reference core.clj:106-108,118,121-122).
"""
from __future__ import annotations

from typing import Callable, Tuple

from . import abi


def shard_params(base: abi.rt_params, rank: int, world: int, mode: str = "tile") -> abi.rt_params:
    """The rt_params of one rank for a job described by `base` (which covers the whole job)."""
    p = abi.rt_params()
    for name, _ in abi.rt_params._fields_:
        setattr(p, name, getattr(base, name)) if name != "background" else None
    p.background[:] = list(base.background)
    if mode == "tile":
        p.shard_index, p.shard_count = rank, world
    elif mode == "sample":
        lo = (base.spp * rank) // world
        hi = (base.spp * (rank + 1)) // world
        p.sample_offset = base.sample_offset + lo
        p.spp = hi - lo
        p.shard_index, p.shard_count = 0, 1
    else:
        raise ValueError(f"unknown split {mode!r}")
    return p


def render_distributed(render_fn: Callable, base: abi.rt_params, rank: int, world: int, mode: str = "tile",
                       reduce_fn: Callable = None):
    """render_fn(params) -> a tensor holding this rank's full-frame sums (zeros where it rendered
    nothing); reduce_fn(tensor) sums it onto rank 0 in place. Returns the tensor (complete on
    rank 0). With world == 1 no collective is issued."""
    p = shard_params(base, rank, world, mode)
    if p.spp <= 0:  # more ranks than samples in "sample" mode: this rank contributes zeros
        p.spp = 1
        acc = render_fn(p)
        acc.zero_()
    else:
        acc = render_fn(p)
    if world > 1:
        if reduce_fn is None:
            import torch.distributed as dist

            def reduce_fn(t):
                dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        reduce_fn(acc)
    return acc
