import sys, os, ctypes
sys.path.insert(0, "/root/repo")
import torch
from pevae_amd import ops, _lib
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
def rnd(m, k): return (torch.rand((m, k), device=dev, generator=gen) * 2 - 1).to(torch.bfloat16)
lib = _lib.load()
for (M, N, K) in ((20004, 512, 12800), (12800, 512, 20032), (1536, 512, 12800), (1536, 128, 12800)):
    A, B = rnd(M, K), rnd(N, K)
    out = torch.zeros((M, N), device=dev)
    for splits in (1, 2, 3, 4, 5, 6, 8):
        nk = K // 64
        per = -(-nk // splits)
        if -(-nk // per) != splits: continue
        ws = torch.empty(lib.pv_gemm_splitk_ws_bytes(M, N, splits), dtype=torch.uint8, device=dev)
        def f():
            if splits == 1:
                ops.gemm_nt(A, K, B, K, M, N, K, out=out, accumulate=True)
            else:
                _lib.call("pv_gemm_bf16_nt_splitk", A.data_ptr(), K, B.data_ptr(), K, M, N, K, splits, ws.data_ptr(), out.data_ptr(), N, 1, None, None, torch.cuda.current_stream().cuda_stream)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        print("%d,%d,%d splits %d: %.1f us  %.0f TF" % (M, N, K, splits, us, 2.0 * M * N * K / us / 1e6))
