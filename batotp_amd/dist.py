"""Multi-GPU plumbing: independent paths are sharded across ranks (one process per GPU); the only
communication is the final gather of the per-path result table (RCCL all_gather over xGMI on GPUs,
gloo on CPU for the tests).  There is no collective inside the hot path: paths never exchange data
(SURVEY.md 8e)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import capi


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous block [lo, hi) of the n_total paths owned by `rank` (first ranks take the remainder)"""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_results(local: np.ndarray, device=None) -> np.ndarray:
    """all_gather the per-path result table (capi.RESULT_DTYPE rows) of every rank, in rank order.

    Rows are fixed-size (64 B), so a padded all_gather of uint8 blocks plus the row counts is
    enough; the tables are tiny (latency-bound), no ring tuning is involved."""
    import torch
    import torch.distributed as dist

    assert local.dtype == capi.RESULT_DTYPE
    if not (dist.is_available() and dist.is_initialized()):
        return local.copy()
    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = max(counts) * capi.RESULT_DTYPE.itemsize
    buf = torch.zeros(max(width, 1), dtype=torch.uint8, device=dev)
    raw = torch.from_numpy(np.frombuffer(local.tobytes(), dtype=np.uint8).copy())
    buf[: raw.numel()] = raw.to(dev)
    parts = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    rows: List[np.ndarray] = []
    for c, t in zip(counts, parts):
        b = t[: c * capi.RESULT_DTYPE.itemsize].cpu().numpy().tobytes()
        rows.append(np.frombuffer(b, dtype=capi.RESULT_DTYPE))
    return np.concatenate(rows) if rows else local.copy()
