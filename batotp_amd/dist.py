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


def gather_curves(batch: "capi.Batch", which: int, device=None, root: int = 0, on_device: bool = False):
    """The integrated curves (traj.sMVC / traj.sdot after sweep, reference ba.cpp:1154-1190) of every rank's paths on rank
    `root`, in rank order: list of (s, sdot) float64 arrays there, None elsewhere.

    The curves have different lengths and RCCL has no gatherv, so: (1) all_gather of the per-path point counts (padded to
    the largest shard), (2) every rank packs its curves into one contiguous (s, sdot)-pair buffer on ITS device
    (batotp_hip_pack_curves), (3) one grouped send/recv moves the buffers device-to-device to the root -- direct peer-to-root
    transfers over the xGMI links of the root, no ring.  With `device` = None (gloo, CPU tests) the same code moves host
    tensors (the checker library's "device" memory is host memory).  on_device = True leaves the result where it arrived: the
    root gets (list of per-rank (points, 2) tensors on its device, list of per-rank point-count arrays) instead of host arrays.
    A rank whose share is empty passes batch = None: it takes part in the collectives and sends nothing."""
    import torch
    import torch.distributed as dist

    if batch is None:
        counts_local = np.zeros(0, dtype=np.int64)
    else:
        res = batch.results()
        counts_local = np.ascontiguousarray(res["n_fwd"] if which == 1 else res["n_rev"], dtype=np.int64)
    dev = device if device is not None else torch.device("cpu")
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if distributed else 1
    rank = dist.get_rank() if distributed else 0

    total = int(counts_local.sum())
    send = torch.empty((max(total, 1), 2), dtype=torch.float64, device=dev)
    if total:
        got = batch.pack_curves(which, 0, batch.n_paths, send.data_ptr(), total)
        assert got == total, (got, total)

    def split(buf, counts):
        host = buf.cpu().numpy()
        out, at = [], 0
        for c in counts:
            c = int(c)
            out.append((host[at:at + c, 0].copy(), host[at:at + c, 1].copy()))
            at += c
        return out

    if world == 1:
        return ([send[:total]], [counts_local]) if on_device else split(send[:total], counts_local)

    # (1) size exchange
    n_local = torch.tensor([counts_local.shape[0]], dtype=torch.int64, device=dev)
    n_all = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(n_all, n_local)
    n_all = [int(t.item()) for t in n_all]
    width = max(max(n_all), 1)
    pad = torch.zeros(width, dtype=torch.int64, device=dev)
    pad[: counts_local.shape[0]] = torch.from_numpy(counts_local).to(dev)
    all_counts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(all_counts, pad)
    counts = [t[:n].cpu().numpy() for t, n in zip(all_counts, n_all)]
    totals = [int(c.sum()) for c in counts]

    # (2) + (3) grouped point-to-point transfers to the root
    ops, recv = [], {}
    if rank == root:
        for r in range(world):
            if r != root and totals[r] > 0:
                recv[r] = torch.empty((totals[r], 2), dtype=torch.float64, device=dev)
                ops.append(dist.P2POp(dist.irecv, recv[r], r))
    elif total > 0:
        ops.append(dist.P2POp(dist.isend, send[:total], root))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != root:
        return None
    bufs = [send[:total] if r == root else recv.get(r, torch.empty((0, 2), dtype=torch.float64, device=dev)) for r in range(world)]
    if on_device:
        return bufs, counts
    out = []
    for r in range(world):
        out.extend(split(bufs[r], counts[r]))
    return out
