"""Data-parallel gradient exchange (pevae_amd/dp.py) with world_size 2 on the gloo backend (CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, bucket_bytes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pevae_amd import dp
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    params = torch.full((n,), float(rank))
    dp.broadcast_parameters(params, src=0)
    red = dp.FlatGradReducer(flat, world, bucket_bytes=bucket_bytes)
    red.reduce()
    torch.save((flat, params), os.path.join(out, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bucket", [(1000, 1 << 20), (70001, 64 << 10)])
def test_flat_grad_reducer_world2(tmp_path, n, bucket):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, bucket, str(tmp_path)), nprocs=2, join=True)
    want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(2)) / 2
    for r in range(2):
        flat, params = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r), weights_only=True)
        torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-6)
        assert torch.equal(params, torch.zeros(n))          # rank 0's parameters everywhere


def test_bucket_bounds_cover_exactly():
    from pevae_amd.dp import bucket_bounds
    for n, b in [(10, 3), (9, 3), (1, 5), (0, 4)]:
        bounds = bucket_bounds(n, b)
        assert [s for s, _ in bounds] == list(range(0, n, b))
        assert sum(e - s for s, e in bounds) == n


def _sparse_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pevae_amd import dp
    dp.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(200 + rank)
    # layout: dense 37 | table A 50 x 8 | dense 1001 | table B 20 x 8 | dense 5
    offA, offB, n = 37, 37 + 400 + 1001, 37 + 400 + 1001 + 160 + 5
    flat = torch.randn(n, generator=g)
    ids = [torch.randint(0, 50, (12,), generator=g), torch.randint(0, 20, (12,), generator=g)]
    ids[0][3] = ids[0][7] = ids[0][0]                      # duplicates inside the local batch
    for off, rows, idv in ((offA, 50, ids[0]), (offB, 20, ids[1])):
        t = flat[off:off + rows * 8].view(rows, 8)
        mask = torch.zeros(rows, dtype=torch.bool)
        mask[idv] = True
        t[~mask] = 0.0                                     # an embedding table's gradient: looked-up rows only
    dense_copy = flat.clone()
    red = dp.FlatGradReducer(flat, world, bucket_bytes=1 << 10, sparse_tables=[(offB, 20, 8), (offA, 50, 8)])
    red.reduce(ids)                                        # ids in table (offset) order: A, B
    torch.save((flat, dense_copy), os.path.join(out, "s%d.pt" % rank))
    dist.destroy_process_group()


def test_sparse_table_exchange_matches_dense_average(tmp_path):
    """User / item tables exchanged as (ids, rows) all-gathers give what a dense all-reduce of the table gives."""
    port = _free_port()
    mp.spawn(_sparse_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r), weights_only=True) for r in range(2)]
    want = (outs[0][1] + outs[1][1]) / 2
    for flat, _ in outs:
        torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(outs[0][0], outs[1][0])             # bit-identical replicas


def _split_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pevae_amd import dp
    dp.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(300 + rank)
    # layout: dense 100 | table 30 x 4 | dense 5000 with the early range [700, 3100) inside it | dense tail
    off, n = 100, 100 + 120 + 5000
    early = (700, 3100)
    grad = torch.randn(n, generator=g)
    ids = torch.randint(0, 30, (9,), generator=g)
    t = grad[off:off + 120].view(30, 4)
    mask = torch.zeros(30, dtype=torch.bool)
    mask[ids] = True
    t[~mask] = 0.0
    flat = torch.zeros(n)
    red = dp.FlatGradReducer(flat, world, bucket_bytes=1 << 10, sparse_tables=[(off, 30, 4)], average=False,
                             early_ranges=[early])
    assert sum(e - s for s, e in red.early_bounds) == early[1] - early[0]
    # the split schedule of bench.py: the early stretch is final first and starts travelling; the rest of the "backward"
    # fills the other stretches while it is in flight; finish() exchanges those and waits for everything
    flat[early[0]:early[1]] = grad[early[0]:early[1]]
    red.start()
    keep = flat[early[0]:early[1]].clone()
    flat[:early[0]] = grad[:early[0]]
    flat[early[1]:] = grad[early[1]:]
    red.finish([ids])
    torch.save((flat, grad), os.path.join(out, "p%d.pt" % rank))
    dist.destroy_process_group()


def test_split_schedule_sum_matches_dense_and_replicas_identical(tmp_path):
    """start() / finish() with an early range (the overlap schedule) and average=False (1/world folded into the optimizer):
    every rank ends with the SUM over ranks, bit-identical across replicas."""
    port = _free_port()
    mp.spawn(_split_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "p%d.pt" % r), weights_only=True) for r in range(2)]
    want = outs[0][1] + outs[1][1]
    for flat, _ in outs:
        torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(outs[0][0], outs[1][0])


def _three_stage_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pevae_amd import dp
    dp.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(400 + rank)
    # layout: dense 100 | table 30 x 4 | dense 5000 | dense tail; early [700, 3100); mid = [0, 100) + [220, 700) + [4000, 5220)
    off, n = 100, 100 + 120 + 5000
    early, mid = (700, 3100), [(0, 100), (220, 700), (4000, n)]
    grad = torch.randn(n, generator=g)
    ids = torch.randint(0, 30, (9,), generator=g)
    t = grad[off:off + 120].view(30, 4)
    mask = torch.zeros(30, dtype=torch.bool)
    mask[ids] = True
    t[~mask] = 0.0
    flat = torch.zeros(n)
    red = dp.FlatGradReducer(flat, world, bucket_bytes=1 << 10, sparse_tables=[(off, 30, 4)], average=False,
                             early_ranges=[early], mid_ranges=mid)
    assert sum(e - s for s, e in red.mid_bounds) == sum(b - a for a, b in mid)
    assert sum(e - s for s, e in red.bounds) == 4000 - 3100
    # bench.py's three-stage schedule: the early stretch, then the mid stretches and the sparse table, then the rest
    flat[early[0]:early[1]] = grad[early[0]:early[1]]
    red.start()
    for a, b in mid:
        flat[a:b] = grad[a:b]
    flat[off:off + 120] = grad[off:off + 120]
    red.start_mid([ids])
    flat[3100:4000] = grad[3100:4000]
    red.finish()
    torch.save((flat, grad), os.path.join(out, "p%d.pt" % rank))
    dist.destroy_process_group()


def test_three_stage_schedule_sum_matches_dense_and_replicas_identical(tmp_path):
    """start() / start_mid(ids) / finish(): early, mid (with the sparse exchange) and final stretches each travel once;
    every rank ends with the SUM over ranks, bit-identical across replicas.  Overlapping stages are rejected."""
    port = _free_port()
    mp.spawn(_three_stage_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "p%d.pt" % r), weights_only=True) for r in range(2)]
    want = outs[0][1] + outs[1][1]
    for flat, _ in outs:
        torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-6)
    assert torch.equal(outs[0][0], outs[1][0])
    from pevae_amd import dp
    with pytest.raises(ValueError):
        dp.FlatGradReducer(torch.zeros(100), 2, early_ranges=[(0, 50)], mid_ranges=[(40, 60)])


# ---- launch layout: bench.py --gpus N starts its own ranks, and a layout that cannot run fails before it can hang ----------
def test_self_launch_command_decides_before_any_gpu_call():
    from pevae_amd import dp
    argv = ["--gpus", "4", "--steps", "5"]
    assert dp.self_launch_command(1, "bench.py", argv, {}, 8) is None                  # N = 1: this process is the run
    assert dp.self_launch_command(4, "bench.py", argv, {"WORLD_SIZE": "4"}, 8) is None  # already a rank of torchrun
    cmd = dp.self_launch_command(4, "/x/bench.py", argv, {}, 8, port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999" and cmd[-5:] == ["/x/bench.py"] + argv
    with pytest.raises(dp.LaunchError, match="shows 1 GPU"):                               # never a silent one-GPU run
        dp.self_launch_command(2, "bench.py", argv, {}, 1)
    port = int(dp.self_launch_command(2, "bench.py", argv, {}, 2)[9])                     # a free port is chosen when none is given
    assert 1024 < port < 65536


def test_init_from_env_rejects_a_world_size_mismatch(monkeypatch):
    from pevae_amd import dp
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(dp.LaunchError, match="WORLD_SIZE = 1 but 8 ranks"):
        dp.init_from_env(backend="gloo", expect_world=8)
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(dp.LaunchError, match="WORLD_SIZE = 4 but 2 ranks"):
        dp.init_from_env(backend="gloo", expect_world=2)


def _layout_worker(rank, world, port, out, shared):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from pevae_amd import dp
    dp.init_from_env(backend="gloo", expect_world=world)
    ident = "pci 0000:05:00" if shared else "pci 0000:%02x:00" % (5 + rank)
    try:
        got = dp.check_one_rank_per_device(ident)
        msg = "ok " + "|".join(got)
    except dp.LaunchError as err:
        msg = "refused: %s" % err
    with open(os.path.join(out, "l%d.txt" % rank), "w") as fh:
        fh.write(msg)
    dist.destroy_process_group()


@pytest.mark.parametrize("shared", [False, True])
def test_two_ranks_on_one_device_are_refused_on_every_rank(tmp_path, shared):
    """The layout that hung round 2's rehearsal (two ranks, one GPU) now ends with a message on every rank."""
    port = _free_port()
    mp.spawn(_layout_worker, args=(2, port, str(tmp_path), shared), nprocs=2, join=True)
    for r in range(2):
        msg = open(os.path.join(str(tmp_path), "l%d.txt" % r)).read()
        if shared:
            assert msg.startswith("refused") and "ranks 0,1 share pci 0000:05:00" in msg
        else:
            assert msg == "ok pci 0000:05:00|pci 0000:06:00"


def test_bench_gpus_n_without_enough_gpus_exits_nonzero_with_the_reason():
    """`python bench.py --gpus 2` on a node with fewer GPUs (here: none) must not fall back to one GPU."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "one rank per GPU is required" in res.stderr and res.stdout.strip() == ""
    env["WORLD_SIZE"] = "4"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode == 2 and "WORLD_SIZE = 4 but 2 ranks" in res.stderr


def _cons_loss(v1, v2):
    """DLS_Auto.cons_loss (dialog_models.py:3266-3272) in plain torch."""
    import torch.nn.functional as F
    v1 = v1 / torch.norm(v1, dim=1).unsqueeze(-1)
    v2 = v2 / torch.norm(v2, dim=1).unsqueeze(-1)
    sc = v1 @ v2.t() * 1.1
    t = torch.arange(sc.shape[0])
    return (F.nll_loss(F.log_softmax(sc, 1), t) + F.nll_loss(F.log_softmax(sc, 0), t)) / 2


def _coupled_losses(x, w1, w2, perm, gather=None, rank=0):
    """A cons_loss over the batch plus a hard-negative term that reads ANOTHER sample's row (the DeN_Auto pattern,
    dialog_models.py:2722-2727) plus a per-sample term; `gather` makes the first two span the global batch."""
    a, b = torch.tanh(x @ w1), torch.tanh(x @ w2)
    if gather is None:
        ga, gb, pool = a, b, b
    else:
        ga, gb, pool = gather(a, replicated=True), gather(b, replicated=True), gather(b)
        perm = perm[rank * x.shape[0]:(rank + 1) * x.shape[0]]
    neg = torch.sigmoid((a * pool[perm]).sum(-1))
    return _cons_loss(ga, gb) - torch.mean(torch.log(1 - neg + 1e-6)) + a.pow(2).sum(-1).mean()


def _global_neg_inputs(B, world):
    g = torch.Generator().manual_seed(77)
    x = torch.randn(world * B, 12, generator=g)
    w1, w2 = torch.randn(12, 8, generator=g) * 0.5, torch.randn(12, 8, generator=g) * 0.5
    perm = torch.randint(0, world * B, (world * B,), generator=g)
    return x, w1, w2, perm


def _global_neg_worker(rank, world, port, B, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    from pevae_amd import dp
    dp.init_from_env(backend="gloo")
    assert dp.global_rows(SimpleNamespace(global_negatives=False)) == (None, 0)           # the default: per shard
    gather, r = dp.global_rows(SimpleNamespace(global_negatives=True))
    assert r == rank and gather.world == world
    x, w1, w2, perm = _global_neg_inputs(B, world)
    w1.requires_grad_(True), w2.requires_grad_(True)
    loss = _coupled_losses(x[rank * B:(rank + 1) * B], w1, w2, perm, gather, rank)
    loss.backward()
    flat = torch.cat([w1.grad.reshape(-1), w2.grad.reshape(-1)])
    dp.FlatGradReducer(flat, world).reduce()                     # the step's own exchange: mean over ranks
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    torch.save((flat, lsum / world), os.path.join(out, "g%d.pt" % rank))
    dist.destroy_process_group()


def test_global_negatives_equal_the_single_process_result_at_twice_the_batch(tmp_path):
    """config.global_negatives: two ranks of B rows with gathered rows give the loss and the parameter
    gradient of ONE process at 2B rows -- for a loss every rank replicates (cons_loss) and for one whose negatives are
    other ranks' rows (the gradient travels back through the gather)."""
    B, world = 6, 2
    port = _free_port()
    mp.spawn(_global_neg_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    x, w1, w2, perm = _global_neg_inputs(B, world)
    w1.requires_grad_(True), w2.requires_grad_(True)
    want = _coupled_losses(x, w1, w2, perm)
    want.backward()
    want_flat = torch.cat([w1.grad.reshape(-1), w2.grad.reshape(-1)])
    for r in range(world):
        flat, loss = torch.load(os.path.join(str(tmp_path), "g%d.pt" % r), weights_only=True)
        torch.testing.assert_close(flat, want_flat, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(loss, want.detach(), rtol=1e-6, atol=1e-6)
    # and the per-shard default is a different loss: the test would notice a gather that does nothing
    local = sum(_coupled_losses(x[r * B:(r + 1) * B], w1, w2, perm[r * B:(r + 1) * B] % B) for r in range(world)) / world
    assert abs(float(local.detach()) - float(want.detach())) > 1e-3
