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


def gather_visual_tokens(local: torch.Tensor, n_items: int, group=None, collective=None) -> torch.Tensor:
    """All-gather the per-rank `[per, tokens, hidden]` blocks into `[n_items, tokens, hidden]` on every rank (padding dropped).
    `collective(full, local)` (optional) replaces torch.distributed's all-gather: RowGather passes the C-ABI ncclAllGather here, so both forms share this index logic."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and collective is None:
        return local[:n_items]
    full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if collective is not None:
        collective(full, local.contiguous())
    else:
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full[:n_items]


def pass_groups(per: int, pass_images: int):
    """[(first, count)] of the tower passes over a rank's `per` images, `pass_images` at a time (the last pass may be short)."""
    g = max(1, int(pass_images))
    return [(a, min(g, per - a)) for a in range(0, per, g)]


def sharded_vision(run_vision: Callable[[torch.Tensor], torch.Tensor], batch: torch.Tensor, group=None, pass_images: int = 0, collective=None) -> torch.Tensor:
    """run_vision maps `[n, ...pixels]` -> `[n, tokens, hidden]` on this rank's device; returns the full `[B, tokens, hidden]`.

    pass_images = 0: the rank's whole slice in one call, ONE all-gather at the end.
    pass_images = g > 0: the slice goes through the tower g images at a time and the all-gather of pass i is issued (async) as soon as its rows exist, so it travels
    over xGMI while the tower works on pass i + 1; a pass's gather returns `[world, g, tokens, hidden]` and image `r * per + a + j` of the batch is row (r, j) of the
    pass that started at a -- the reassembly below.  `collective(full, local)`, when given, is used instead (synchronously: the C-ABI form runs on the engine's stream)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_items = batch.shape[0]
    mine = local_slice(batch, rank, world)
    if pass_images <= 0 or world == 1:
        return gather_visual_tokens(run_vision(mine), n_items, group, collective)
    per = mine.shape[0]
    pending = []
    for a, cnt in pass_groups(per, pass_images):
        loc = run_vision(mine[a:a + cnt]).contiguous()
        full = torch.empty((world * cnt,) + tuple(loc.shape[1:]), dtype=loc.dtype, device=loc.device)
        if collective is not None:
            collective(full, loc)
            work = None
        else:
            work = dist.all_gather_into_tensor(full, loc, group=group, async_op=True)
        pending.append((a, cnt, loc, full, work))
    out = None
    for a, cnt, loc, full, work in pending:
        if work is not None:
            work.wait()
        if out is None:
            out = torch.empty((world * per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
        out.view(world, per, *full.shape[1:])[:, a:a + cnt] = full.view(world, cnt, *full.shape[1:])
    return out[:n_items]


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

    def collective(self, stream: int = 0):
        """`collective(full, local)` for gather_visual_tokens / sharded_vision: one ncclAllGather through the C ABI on `stream`, then a sync (the caller reads `full`
        on its own stream)."""
        L = self._L

        def run(full: torch.Tensor, local: torch.Tensor) -> None:
            rows, cols = local.shape[0] * local.shape[1], local.shape[2]
            assert full.dtype == torch.float32 and local.dtype == torch.float32 and full.numel() == self.world * local.numel()
            L.check(L.load().mllm_hip_all_gather_rows(self._comm, L.vp(local), L.vp(full), L.i64(rows), C.c_int(cols), L.vp(stream)), "all_gather_rows")
            L.check(L.load().mllm_hip_sync(L.vp(stream)), "sync")
        return run

    def gather(self, local: torch.Tensor, n_items: int, stream: int = 0) -> torch.Tensor:
        """local `[per, tokens, hidden]` fp32 on this rank's device -> `[n_items, tokens, hidden]` on every rank, rank order, padding dropped (the index logic is
        gather_visual_tokens', shared with the torch.distributed form)."""
        return gather_visual_tokens(local.contiguous(), n_items, collective=self.collective(stream)) if self.world > 1 else local[:n_items]

    def close(self):
        if self._comm:
            self._L.load().mllm_hip_comm_destroy(self._comm)
            self._comm = C.c_void_p()
