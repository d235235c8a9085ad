"""Multi-GPU part of the hot path: the vision prefill shards by image, the LLM does not shard ("replicas only").

Images are independent units (one `visual(...)` call per image, no cross-image attention: modeling_qwen2_vl.hpp:177-190), so a
batch of B images is split into contiguous, equally sized slices, one per rank (the last slices are padded by repeating the
last image so that every rank sends the same count), each rank runs the ViT on its slice, and ONE all-gather reassembles the
`[B, tokens, hidden]` visual-token block on every rank.  The collective is `torch.distributed.all_gather_into_tensor`: RCCL over
xGMI with the "nccl" backend on GPUs, gloo on CPU in the tests.  Per-rank message: 256 x 1536 fp32 = 1.5 MiB per 448x448 image,
latency-bound on 7 x 153 GB/s links, <1 % of a ViT forward -- issued on the compute stream, overlapped with nothing.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(start, valid_count, padded_count) of the slice of `rank`; every rank gets padded_count = ceil(n/world) items."""
    per = (n_items + world - 1) // world
    start = min(rank * per, n_items)
    valid = max(0, min(per, n_items - start))
    return start, valid, per


def local_slice(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's images, padded (by repeating the batch's last image) to the common per-rank count."""
    start, valid, per = shard_range(batch.shape[0], rank, world)
    sl = batch[start:start + valid]
    if valid < per:
        pad = batch[-1:].expand(per - valid, *batch.shape[1:])
        sl = torch.cat([sl, pad], dim=0)
    return sl.contiguous()


def gather_visual_tokens(local: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """All-gather the per-rank `[per, tokens, hidden]` blocks into `[n_items, tokens, hidden]` on every rank (padding dropped)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:n_items]
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:n_items]


def sharded_vision(run_vision: Callable[[torch.Tensor], torch.Tensor], batch: torch.Tensor, group=None) -> torch.Tensor:
    """run_vision maps `[n, ...pixels]` -> `[n, tokens, hidden]` on this rank's device; returns the full `[B, tokens, hidden]`."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return gather_visual_tokens(run_vision(local_slice(batch, rank, world)), batch.shape[0], group)


class RowGather:
    """The same exchange through the C ABI (mllm_hip_comm_* / mllm_hip_all_gather_rows: one ncclAllGather of RCCL on a given stream) -- the form a C++ host
    (the reference-side HIPBackend) uses, with no torch type crossing the boundary.  The 128-byte unique id travels from rank 0 to the other
    ranks by whatever bootstrap the host has; here torch.distributed's object broadcast (only the id: the tokens never touch it)."""

    def __init__(self, world: int, rank: int, group=None):
        from . import lib as L
        self._L, self.world, self.rank = L, world, rank
        idbuf = (C.c_uint8 * 128)()
        if rank == 0:
            L.check(L.load().mllm_hip_comm_unique_id(idbuf), "comm_unique_id")
        box = [bytes(idbuf)]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self._comm = C.c_void_p()
        L.check(L.load().mllm_hip_comm_create(box[0], C.c_int(world), C.c_int(rank), C.byref(self._comm)), "comm_create")

    def gather(self, local: torch.Tensor, n_items: int, stream: int = 0) -> torch.Tensor:
        """local `[per, tokens, hidden]` fp32 on this rank's device -> `[n_items, tokens, hidden]` on every rank, rank order, padding dropped."""
        L = self._L
        local = local.contiguous()
        per = local.shape[0]
        rows, cols = per * local.shape[1], local.shape[2]
        full = torch.empty((self.world * per,) + tuple(local.shape[1:]), dtype=torch.float32, device=local.device)
        L.check(L.load().mllm_hip_all_gather_rows(self._comm, L.vp(local), L.vp(full), L.i64(rows), C.c_int(cols), L.vp(stream)), "all_gather_rows")
        L.check(L.load().mllm_hip_sync(L.vp(stream)), "sync")       # the caller reads `full` on its own stream
        return full[:n_items]

    def close(self):
        if self._comm:
            self._L.load().mllm_hip_comm_destroy(self._comm)
            self._comm = C.c_void_p()
