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


def _fake_embed4(rows):
    rows = torch.as_tensor(rows, dtype=torch.float32)
    return torch.stack([rows, torch.sin(rows), rows * 0.5 + 1.0, torch.cos(rows * 3.0)], dim=1)


def _worker8(rank, world, port, n_items, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def embed(items):
            seen.append((int(items[0]), int(items[-1]) + 1) if len(items) else (0, 0))
            return _fake_embed4(items)
        full = svdist.sharded_embed(embed, torch.arange(n_items))
        want = _fake_embed4(torch.arange(n_items))
        ok = full.shape == want.shape and bool(torch.equal(full, want))
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([int(ok), seen[0][0], seen[0][1], full.shape[0]]))
    finally:
        dist.destroy_process_group()


def test_eight_ranks_at_the_corpus_size(tmp_path):
    """BASELINE config 5's exchange in full size on CPU: 148 642 items over EIGHT gloo ranks (shards of 18 581 x 7 + 18 575:
    uneven, the last one zero-padded for the gather), 4-float rows: the gathered matrix equals the single-process one on
    every rank, and every rank embedded exactly its contiguous range."""
    n_items, world = 148642, 8
    port = _free_port()
    mp.spawn(_worker8, args=(world, port, n_items, str(tmp_path)), nprocs=world, join=True)
    at = 0
    for r in range(world):
        ok, lo, hi, rows = (int(v) for v in np.load(tmp_path / f"ok{r}.npy"))
        assert ok == 1 and rows == n_items, r
        assert (lo, hi) == svdist.shard_bounds(n_items, world, r) and lo == at
        at = hi
    assert at == n_items and svdist.shard_bounds(n_items, world, 7) == (130067, 148642)


def test_single_process_passthrough():
    x = _fake_embed(torch.arange(5))
    assert torch.equal(svdist.all_gather_embeddings(x, 5), x)
    assert torch.equal(svdist.sharded_embed(_fake_embed, torch.arange(5)), x)


def _train_worker(rank, world, port, out_dir):
    """Two ranks, each with half of the pair batch; the averaged gradient must equal the gradient of
    the mean of the two local losses (what one process would compute on the whole batch)."""
    from speaker_verification_amd.model import seeded_model
    from speaker_verification_amd.train_siamese import allreduce_gradients, make_criterion
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = seeded_model(9, n_labels=4)
        crit = make_criterion(0.001, 2.0)
        g = torch.Generator().manual_seed(100)
        a = torch.randn(4, 1, 20, 80, 40, generator=g)
        b = torch.randn(4, 1, 20, 80, 40, generator=g)
        y = torch.tensor([1.0, 0.0, 0.0, 1.0])
        lo, hi = rank * 2, rank * 2 + 2
        model.eval()                                    # BN in eval mode: shard-independent statistics
        loss = crit(model, y[lo:hi], model(a[lo:hi], development=False), model(b[lo:hi], development=False))
        loss.backward()
        allreduce_gradients(model)
        flat = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
        np.save(os.path.join(out_dir, f"g{rank}.npy"), flat.numpy())
        if rank == 0:
            ref = seeded_model(9, n_labels=4).eval()
            l0 = crit(ref, y[0:2], ref(a[0:2], development=False), ref(b[0:2], development=False))
            l1 = crit(ref, y[2:4], ref(a[2:4], development=False), ref(b[2:4], development=False))
            (0.5 * (l0 + l1)).backward()
            want = torch.cat([p.grad.reshape(-1) for p in ref.parameters() if p.grad is not None])
            np.save(os.path.join(out_dir, "want.npy"), want.numpy())
    finally:
        dist.destroy_process_group()


def test_siamese_gradient_allreduce(tmp_path):
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1, want = (np.load(tmp_path / f) for f in ("g0.npy", "g1.npy", "want.npy"))
    np.testing.assert_array_equal(g0, g1)
    np.testing.assert_allclose(g0, want, rtol=1e-4, atol=1e-6)


def test_siamese_train_step_single_process():
    from speaker_verification_amd.model import seeded_model
    from speaker_verification_amd.train_siamese import make_criterion, siamese_train_step
    model = seeded_model(2, n_labels=4)
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(4, 1, 20, 80, 40, generator=g), torch.randn(4, 1, 20, 80, 40, generator=g)
    y = torch.tensor([1.0, 0.0, 1.0, 0.0])
    before = model.FC5.weight.detach().clone()
    l1 = siamese_train_step(model, make_criterion(), opt, a, b, y)
    assert np.isfinite(l1) and not torch.equal(before, model.FC5.weight)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE starts two fresh rank processes itself (no torchrun),
    relays rank 0's one JSON line and reports what torch.distributed saw (--selftest: gloo, no GPU)."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in (2, 1):
        proc = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--selftest"],
                              env=env, stdout=subprocess.PIPE, timeout=300)
        assert proc.returncode == 0
        lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["ranks_seen"] == n and rec["n_gpus"] == n and rec["gathered_ok"] and rec["max_over_ranks"] == n
        assert rec["backend"] == ("gloo" if n > 1 else None)


def test_bench_selftest_under_torch_distributed_run():
    """The driver's launch form for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`): every process is one rank and reads RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* from the environment -- no self-launch; rank 0 alone prints the one JSON line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "4", "--selftest"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, proc.stdout.decode()
    rec = json.loads(lines[0])
    assert rec["ranks_seen"] == 4 and rec["n_gpus"] == 4 and rec["gathered_ok"] and rec["max_over_ranks"] == 4 and rec["backend"] == "gloo"


def test_bench_launcher_reports_a_failed_rank(tmp_path):
    """A rank that dies makes the launcher exit non-zero (it never re-execs itself: plain child processes)."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest", "--steps", "x"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert proc.returncode != 0


@pytest.mark.parametrize("dead", [1, 0])
def test_bench_launcher_fails_fast_when_one_rank_dies(dead):
    """ONE rank exits before `init_process_group` (a bad device index, a missing MIOpen db, OOM ...)
    while the other sits in the rendezvous.  The launcher must notice the dead child, terminate the waiting one, print
    the dead rank's stderr and return non-zero in seconds -- not after torch's 10 / 30 minute timeout."""
    import subprocess
    import sys
    import time
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SVK_BENCH_FAIL_RANK"] = str(dead)
    t0 = time.time()
    proc = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    took = time.time() - t0
    err = proc.stderr.decode()
    assert proc.returncode != 0 and took < 30, (proc.returncode, took, err[-500:])
    assert proc.stdout.decode().strip() == ""                         # no JSON line from a failed job
    assert "rank %d exited with code 3" % dead in err and "SVK_BENCH_FAIL_RANK" in err, err[-800:]


def test_bench_refuses_more_rccl_ranks_than_devices():
    """`--gpus N` over RCCL on a host with fewer devices fails before any GPU call, with the reason (this container has
    no GPU at all: device_count() = 0)."""
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("host has two devices")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    proc = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"],
                          env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert proc.returncode != 0 and time.time() - t0 < 60
    assert "needs 2 devices" in proc.stderr.decode()
