"""N > 1 path of the hot path on CPU: the image-batch shard + single all-gather of the vision prefill, world_size 2 and 3 over gloo
(the GPU runs use the same code with the "nccl" = RCCL backend).  The per-image function is a deterministic stand-in for the ViT:
what is under test is the partition, the padding and the reassembly order, not arithmetic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mllm_amd import parallel


def fake_vision(pix: torch.Tensor) -> torch.Tensor:
    # [n, patches, pe] -> [n, patches/4, 8]: depends on every pixel of the image and on nothing else
    n, p, pe = pix.shape
    return pix.reshape(n, p // 4, 4 * pe)[:, :, :8] * 2.0 + pix.sum(dim=(1, 2), keepdim=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_img, out_dir, pass_images=0, injected=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    batch = torch.randn(n_img, 16, 6)
    calls = []
    collective = None
    if injected:      # the shape of RowGather.collective(): `collective(full, local)` -- here gloo stands in for the C-ABI ncclAllGather, the index logic is the shared one
        def collective(full, local):
            calls.append(tuple(local.shape))
            dist.all_gather_into_tensor(full, local)
    full = parallel.sharded_vision(fake_vision, batch, pass_images=pass_images, collective=collective)
    if injected:
        assert len(calls) == (len(parallel.pass_groups(parallel.shard_range(n_img, rank, world)[2], pass_images)) if pass_images else 1)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_img", [(2, 8), (2, 5), (3, 8), (2, 1)])
def test_sharded_vision_allgather(tmp_path, world, n_img):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_img, str(tmp_path)), nprocs=world, join=True)
    torch.manual_seed(0)
    want = fake_vision(torch.randn(n_img, 16, 6)).numpy()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"rank {r}"


def test_shard_range_partition():
    for n in range(0, 20):
        for w in range(1, 9):
            seen = []
            per = None
            for r in range(w):
                s, v, p = parallel.shard_range(n, r, w)
                per = p if per is None else per
                assert p == per and 0 <= v <= p
                seen += list(range(s, s + v))
            assert seen == list(range(n))


@pytest.mark.parametrize("world,n_img,pass_images,injected", [(2, 8, 2, False), (2, 7, 3, False), (3, 8, 1, False), (2, 8, 0, True), (2, 9, 2, True), (3, 5, 4, True)])
def test_per_pass_all_gather_and_injected_collective(tmp_path, world, n_img, pass_images, injected):
    """The all-gather issued per tower pass (async, overlapped with the next pass) reassembles the same `[B, tokens, hidden]` block as the single gather, for pass sizes
    that do and do not divide the per-rank count, uneven batches included; and the `collective(full, local)` hook RowGather plugs the C-ABI ncclAllGather into goes
    through the very same partition / padding / reassembly code (gloo stands in for RCCL, which needs GPUs)."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_img, str(tmp_path), pass_images, injected), nprocs=world, join=True)
    torch.manual_seed(0)
    want = fake_vision(torch.randn(n_img, 16, 6)).numpy()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == want.shape and np.array_equal(got, want), f"rank {r}"


def test_pass_groups():
    assert parallel.pass_groups(7, 3) == [(0, 3), (3, 3), (6, 1)]
    assert parallel.pass_groups(4, 0) == [(0, 1), (1, 1), (2, 1), (3, 1)]
    assert parallel.pass_groups(4, 9) == [(0, 4)]
