import torch, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import _lib, ops
dev = torch.device("cuda:0")
for W in (1, 2, 4, 8):
    rows, D, n = 30000, 256, 256
    S = n * (D + 1)
    recv = torch.zeros(W * S, device=dev)
    for r in range(W):
        recv[r * S:r * S + n] = torch.randint(0, rows, (n,)).float().to(dev)
        recv[r * S + n:(r + 1) * S] = torch.randn(n * D, device=dev)
    table = torch.zeros(rows, D, device=dev)
    for _ in range(3):
        _lib.call("pv_sparse_merge", ops._p(recv), W, S, n, D, rows, ops._p(table), ops._stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _lib.call("pv_sparse_merge", ops._p(recv), W, S, n, D, rows, ops._p(table), ops._stream())
    e1.record()
    torch.cuda.synchronize()
    print("world %d: pv_sparse_merge %.1f us" % (W, e0.elapsed_time(e1) / 20 * 1e3))
