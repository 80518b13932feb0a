"""Times pv_embedding_sort (the single-workgroup LDS radix sort of csrc/embedding.hip) at the step's lookup sizes."""
import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
from pevae_amd import _lib
dev = torch.device("cuda:0")
for rows, n in ((20004, 12800), (30000, 256), (200001, 12800), (20004, 16384)):
    ids = torch.randint(0, rows, (n,), device=dev)
    D = 100
    grad = torch.randn(n, D, device=dev)
    dt = torch.zeros(rows, D, device=dev)
    nb = _lib.load().pv_embedding_bwd_ws_bytes(n, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: _lib.call("pv_embedding_sort", C.c_void_p(ids.data_ptr()), n, n, n, 0, 1, rows, D, C.c_void_p(ws.data_ptr()), nb, st)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(rows, n, "sort %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
