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
