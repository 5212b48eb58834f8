"""Data-parallel sharding of the hot path across the GPUs of one node.

Frames and skeleton windows are independent units: rank r of R takes the contiguous slice
``[r*n/R, (r+1)*n/R)``, weights and the support set are replicated, and the only collective is ONE
all-gather of the packed per-window record ``[logits(n_classes) | is_true(1) | embed(L*256, optional)]``
(RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" in the CPU tests).
There is no cross-rank arithmetic, so a sharded run is bit-identical to the unsharded one.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice of n units for `rank` (first n % world ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0,{world})")
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_records(logits, is_true, embed=None):
    """[B,n] , [B] , optional [B,L,256] -> [B, n+1(+L*256)] contiguous record tensor."""
    import torch
    parts = [logits, is_true.reshape(-1, 1)]
    if embed is not None:
        parts.append(embed.reshape(embed.shape[0], -1))
    return torch.cat(parts, dim=1).contiguous()


def unpack_records(rec, n_classes: int, seq_len: Optional[int] = None):
    logits = rec[:, :n_classes]
    is_true = rec[:, n_classes]
    embed = None
    if rec.shape[1] > n_classes + 1:
        embed = rec[:, n_classes + 1:].reshape(rec.shape[0], seq_len, -1)
    return logits, is_true, embed


def all_gather_records(rec, counts=None):
    """One all-gather of the per-rank record blocks. Equal shard sizes use
    all_gather_into_tensor (one fused collective); ragged shards are padded to the largest one."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if world == 1 and os.environ.get("ISB_BENCH_FORCE_DIST") != "1":     # (the switch: exercise the collective with one rank)
        return rec
    if rec.is_cuda and dist.get_backend() == "gloo":
        # gloo has no device all-gather: only reached when the N > 1 path is rehearsed on a one-GPU box
        # (bench.py, ISB_BENCH_BACKEND=gloo); RCCL ("nccl") gathers the device tensors directly
        return all_gather_records(rec.cpu(), counts).to(rec.device)
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec)
        return out
    mx = max(counts)
    pad = torch.zeros((mx, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[: rec.shape[0]] = rec
    out = torch.empty((world * mx, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
