"""Main loop of the bf16 GEMM core without an epilogue (diagnostic library built with -DPV_GEMM_DEBUG)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
def rnd(m, k): return (torch.rand((m, k), device=dev, generator=gen) * 2 - 1).to(torch.bfloat16)
sink = torch.zeros(4, device=dev)
f = lib.pv_gemm_bf16_nt_noepi
f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
for M, N, K in ((12800, 20032, 512), (12800, 512, 20032), (20032, 512, 12800), (8192, 8192, 8192)):
    A, B = rnd(M, K), rnd(N, K)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): f(A.data_ptr(), K, B.data_ptr(), K, M, N, K, sink.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f(A.data_ptr(), K, B.data_ptr(), K, M, N, K, sink.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print("%d,%d,%d no-epilogue: %.1f us  %.0f TF" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
