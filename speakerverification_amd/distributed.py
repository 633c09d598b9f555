"""Multi-GPU plumbing: one process per GPU.

The path shards by utterance: rank r embeds the contiguous block [r*ceil(N/W), ...) of the sorted
unique file list, then ONE all-gather of the dense (N/W, nOut) fp32 block per rank assembles the
embedding matrix on every rank (reference: pickled ``all_gather_object`` of dicts,
src/model.py:400-411).  Scoring is then row-sharded by enrol index (``trial_rows_of_rank``) or done on
rank 0 as the reference does.

Two carriers for that one collective:
  * ``LibComm`` — RCCL under the C ABI (``svhip_comm_init`` / ``svhip_allgather_rows``): the all-gather is enqueued on the
    Engine's own HIP stream with raw pointers; torch.distributed (any backend, gloo is enough) only ships the 128-byte
    RCCL id at start-up.  This is what bench.py and ``ModelHandling`` use on GPUs.
  * ``all_gather_rows`` over ``torch.distributed`` (backend "nccl" is RCCL on ROCm, "gloo" on CPU) — kept for hosts that
    already own a process group and for the CPU tests.
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


def trial_rows_of_rank(ia, n_total: int, rank: int, world: int):
    """Row-sharded scoring: rank r scores the trials whose ENROL index falls in its utterance block.
    Returns the positions (into the trial list) this rank owns; over all ranks they partition the list."""
    lo, hi, _ = shard_bounds(n_total, rank, world)
    ia = np.asarray(ia)
    return np.nonzero((ia >= lo) & (ia < hi))[0]


def all_gather_rows(local, n_total: int):
    """local: this rank's (n_local, ...) block in shard_bounds order (torch tensor, CPU for gloo /
    CUDA for nccl).  Returns the (n_total, ...) matrix on every rank with ONE collective."""
    rank, world = rank_world()
    if world == 1 and not is_distributed():
        return local
    lo, hi, per = shard_bounds(n_total, rank, world)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    padded = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return out[:n_total]


class LibComm:
    """RCCL communicator owned by an Engine (the C ABI's svhip_comm_*).  Construction is collective: every rank of the
    torch.distributed group (used once, to broadcast rank 0's RCCL id) must build one on its own Engine."""

    def __init__(self, engine, rank=None, world=None, id_bytes=None):
        if rank is None or world is None:
            rank, world = rank_world()
        if id_bytes is None:
            box = [engine.comm_unique_id() if rank == 0 else None]
            if world > 1:
                if not is_distributed():
                    raise RuntimeError("LibComm needs torch.distributed (any backend) to ship the RCCL id, or id_bytes=")
                dist.broadcast_object_list(box, src=0)
            id_bytes = box[0]
        engine.comm_init(id_bytes, rank, world)
        self.engine, self.rank, self.world = engine, rank, world

    def all_gather_rows(self, local, n_total: int, out=None):
        """local: (n_local, ...) fp32 block (numpy -> numpy, CUDA tensor -> CUDA tensor) in shard_bounds order;
        returns the (n_total, ...) matrix on every rank."""
        lo, hi, per = shard_bounds(n_total, self.rank, self.world)
        assert local.shape[0] == hi - lo, (local.shape, lo, hi)
        tail = tuple(local.shape[1:])
        D = int(np.prod(tail)) if tail else 1
        is_t = torch is not None and isinstance(local, torch.Tensor)
        if hi - lo == per:
            padded = local.reshape(per, D)
        elif is_t:
            padded = torch.zeros((per, D), dtype=torch.float32, device=local.device)
            padded[: hi - lo] = local.reshape(hi - lo, D)
        else:
            padded = np.zeros((per, D), np.float32)
            padded[: hi - lo] = np.asarray(local, np.float32).reshape(hi - lo, D)
        if not is_t:
            padded = np.ascontiguousarray(padded, dtype=np.float32)
        full = self.engine.allgather_rows(padded, out=out)
        return full[:n_total].reshape((n_total,) + tail)
