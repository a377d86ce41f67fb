"""N > 1 path on CPU: two gloo ranks shard a corpus, embed their shards and all-gather.
The embedding function here is the CPU oracle chain (the HIP kernels need a GPU); what is
under test is the sharding / padding / all-gather / scoring logic of
speaker_verification_amd.distributed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speaker_verification_amd import distributed as svdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_embed(rows):
    """Deterministic per-item 'embedding' so every rank can predict the gathered matrix."""
    rows = torch.as_tensor(rows, dtype=torch.float32)
    return torch.stack([torch.sin(rows * (k + 1)) for k in range(8)], dim=1)


def _worker(rank, world, port, n_items, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = torch.arange(n_items)
        full = svdist.sharded_embed(_fake_embed, items)
        lo, hi = svdist.shard_bounds(n_items, world, rank)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"b{rank}.npy"), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [11, 16, 1])
def test_two_rank_all_gather(tmp_path, n_items):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_items, str(tmp_path)), nprocs=2, join=True)
    want = _fake_embed(torch.arange(n_items)).numpy()
    for r in range(2):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    b0, b1 = np.load(tmp_path / "b0.npy"), np.load(tmp_path / "b1.npy")
    assert b0[0] == 0 and b0[1] == b1[0] and b1[1] == n_items


def test_single_process_passthrough():
    x = _fake_embed(torch.arange(5))
    assert torch.equal(svdist.all_gather_embeddings(x, 5), x)
    assert torch.equal(svdist.sharded_embed(_fake_embed, torch.arange(5)), x)
