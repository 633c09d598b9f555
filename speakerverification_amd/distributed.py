"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by utterance: rank r embeds the contiguous block [r*ceil(N/W), ...) of the sorted
unique file list, then ONE all-gather of the dense (N/W, nOut) fp32 block per rank assembles the
embedding matrix on every rank (reference: pickled ``all_gather_object`` of dicts,
src/model.py:400-411).  Scoring is then row-sharded or done on rank 0.
"""
from __future__ import annotations

import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def is_distributed() -> bool:
    return dist is not None and dist.is_available() and dist.is_initialized()


def rank_world():
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world: int):
    """contiguous block partition, last blocks may be short / empty"""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi, per


def all_gather_rows(local, n_total: int):
    """local: this rank's (n_local, ...) block in shard_bounds order (torch tensor, CPU for gloo /
    CUDA for nccl).  Returns the (n_total, ...) matrix on every rank with ONE collective."""
    rank, world = rank_world()
    if world == 1:
        return local
    lo, hi, per = shard_bounds(n_total, rank, world)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return out[:n_total]
