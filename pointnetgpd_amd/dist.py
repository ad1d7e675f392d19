"""Data-parallel sharding of candidates over the GPUs of one node: one process per GPU.

Inference needs no collective: weights are replicated (6.4 MB), the scene cloud is replicated,
every rank scores a contiguous slice of the candidates and keeps (or gathers, 8 bytes per grasp)
its results.  The reference's nn.DataParallel scatter/gather (main_test.py:66-68) is not
reproduced.  The only data-path collective of the framework is the gradient all-reduce of the
training step (pointnetgpd_amd.training).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int):
    """Contiguous, balanced slice [lo, hi) of n items for `rank` (first n % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def broadcast_int(value: int, device=None, src: int = 0) -> int:
    """`value` of rank `src` on every rank (the value itself when no process group is up)."""
    if world()[1] == 1:
        return int(value)
    on_device = dist.get_backend() == "nccl"
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if on_device else "cpu")
    dist.broadcast(t, src=src)
    return int(t.item())


def gather_concat(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank result slices (ragged along dim 0) on every rank, in rank order."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_range(n_total, r, ws) for r in range(ws)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    staged = pad.is_cuda and dist.get_backend() != "nccl"      # a host-memory backend (gloo): stage through the host
    send = pad.cpu() if staged else pad
    bufs = [torch.empty_like(send) for _ in range(ws)]
    dist.all_gather(bufs, send)
    out = torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
    return out.to(local.device) if staged else out


def score_grasps_sharded(model, scene, grasps, limits, repeat: int = 3, seed: int = 0, gather: bool = True):
    """Each rank scores its slice of `grasps` (all ranks pass the full array); with gather=True all
    ranks return the full (label, score, valid) vectors.  Draws are keyed by the GLOBAL grasp id so
    the result does not depend on the number of ranks."""
    from . import crop as _crop
    from .scoring import score_candidates

    rank, ws = world()
    G = grasps.shape[0] if hasattr(grasps, "shape") else len(grasps)
    lo, hi = shard_range(G, rank, ws)
    local = grasps[lo:hi]
    P = model.num_points
    # the device generator keys every draw by the global grasp id (slice offset `lo`)
    pts, valid, _, overflow = _crop.crop_subsample(local, scene, limits, P, repeat=repeat, seed=seed, grasp_offset=lo)
    pred, probs, _ = score_candidates(model, pts)
    label, score = _crop.vote(pred, probs, valid, repeat)
    if int(overflow.item()) != 0:      # a grasp with more points than the one-pass kernel holds: three-call path
        res = _crop.collect_pc(local, scene, limits, return_points=False)
        pts, valid = _crop.subsample(res, P, repeat=repeat, seed=seed, grasp_offset=lo)
        pred, probs, _ = score_candidates(model, pts)
        label, score = _crop.vote(pred, probs, valid, repeat)
    if not gather:
        return label, score, valid
    return gather_concat(label, G), gather_concat(score, G), gather_concat(valid, G)
