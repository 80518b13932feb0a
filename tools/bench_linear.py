"""Micro-benchmark of the f32 MFMA small GEMM (pv_linear_f32) over the pyramid's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import ops
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    return ts[len(ts) // 2] * 1e3
for (M, K, N) in [(256, 512, 512), (256, 512, 256), (256, 768, 512), (256, 64, 128), (256, 32, 32), (256, 1024, 512)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    f = lambda: ops._lin(x, K, 1, w, 1, K, b, y, N, M, N, K)
    g = lambda: ops._lin(dy, N, 1, w, K, 1, None, dx, K, M, K, N)
    h = lambda: ops._lin(dy, 1, N, x, K, 1, None, dw, K, N, K, M)
    print("M=%d K=%d N=%d  fwd %.1f us  dx %.1f us  dW %.1f us" % (M, K, N, t(f), t(g), t(h)), flush=True)
