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
