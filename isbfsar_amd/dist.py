"""Data-parallel sharding of the hot path across the GPUs of one node.

Frames and skeleton windows are independent units: rank r of R takes the contiguous slice
``[r*n/R, (r+1)*n/R)``, weights and the support set are replicated, and the only collective is ONE
all-gather of the packed per-window record ``[logits(n_classes) | is_true(1) | embed(L*256, optional)]``
(on the GPU box: RCCL over xGMI behind the C ABI, isb_dist_* -- class RecordGather; torch.distributed "gloo" as the test
double in the CPU tests).
There is no cross-rank arithmetic, so a sharded run is bit-identical to the unsharded one.
"""
from __future__ import annotations

import ctypes as C
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


class RecordGather:
    """The step's ONE collective behind the C ABI (include/isbfsar.h, isb_dist_*): an RCCL communicator owned by the
    library, ncclAllGather issued on the CURRENT stream -- so it can be captured in the hipGraph of a streaming step.
    The 128-byte RCCL id travels from rank 0 over the launcher's torch.distributed group (any backend); after that the
    data path does not touch torch.distributed any more."""

    def __init__(self, device: int, group=None):
        import torch
        import torch.distributed as dist
        from . import _lib
        self._lib = _lib
        self.device = device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        ident = [None]
        if self.rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(_lib.lib().isb_dist_unique_id(buf), "isb_dist_unique_id")
            ident[0] = bytes(buf.raw)
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().isb_dist_create(ident[0], self.rank, self.world, device, C.byref(self._h)), "isb_dist_create")
        self._torch = torch

    def rccl_ranks(self) -> int:
        """ranks in the communicator as RCCL itself counts them (ncclCommCount)"""
        n = C.c_int32()
        self._lib.check(self._lib.lib().isb_dist_comm_count(self._h, C.byref(n)), "isb_dist_comm_count")
        return int(n.value)

    def all_gather(self, rec):
        """rec [n, w] (equal n on every rank) -> [world * n, w], on the current stream"""
        torch = self._torch
        rec = rec.contiguous()
        out = torch.empty((self.world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        stream = torch.cuda.current_stream(rec.device).cuda_stream
        self._lib.check(self._lib.lib().isb_dist_all_gather(self._h, rec.data_ptr(), out.data_ptr(), rec.numel() * rec.element_size(),
                                                            C.c_void_p(stream)), "isb_dist_all_gather")
        return out

    def all_gather_into(self, rec, out):
        """the same into a caller-owned tensor (hipGraph capture: no allocation inside the captured region)"""
        torch = self._torch
        assert rec.is_contiguous() and out.is_contiguous() and out.numel() == self.world * rec.numel()
        stream = torch.cuda.current_stream(rec.device).cuda_stream
        self._lib.check(self._lib.lib().isb_dist_all_gather(self._h, rec.data_ptr(), out.data_ptr(), rec.numel() * rec.element_size(),
                                                            C.c_void_p(stream)), "isb_dist_all_gather")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lib().isb_dist_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def all_gather_records(rec, counts=None, force: bool = False, gather: Optional[RecordGather] = None):
    """One all-gather of the per-rank record blocks. `gather` (a RecordGather: RCCL behind the C ABI) is the GPU path;
    without it the exchange goes through torch.distributed -- the gloo test double of the CPU tests and of one-GPU
    rehearsals. Equal shard sizes make one fused collective; ragged shards are padded to the largest one.
    force: run the collective even with ONE rank (bench.py's one-rank rehearsal of the N > 1 code path)."""
    import torch
    import torch.distributed as dist
    if gather is not None:
        world = gather.world
        if counts is None or len(set(counts)) == 1:
            return gather.all_gather(rec)
        mx = max(counts)
        pad = torch.zeros((mx, rec.shape[1]), dtype=rec.dtype, device=rec.device)
        pad[: rec.shape[0]] = rec
        out = gather.all_gather(pad)
        return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
    world = dist.get_world_size()
    if world == 1 and not force:
        return rec
    if rec.is_cuda and dist.get_backend() == "gloo":
        # gloo has no device all-gather: only reached when the N > 1 path is rehearsed on a one-GPU box
        # (bench.py, ISB_BENCH_BACKEND=gloo); RCCL ("nccl") gathers the device tensors directly
        return all_gather_records(rec.cpu(), counts, force).to(rec.device)
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
