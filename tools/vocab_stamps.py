"""Diagnostic: where a wave of the fused vocabulary kernels spends its cycles, per section of the item loop (needs the
stamps build: make -C pevae_amd/csrc stamps; PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so python tools/vocab_stamps.py).
Shares only: the stamps drain the LDS queue at every section boundary, so this build is slower than the real one."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pevae_amd import _lib

dev = torch.device("cuda:0")
T, V, H = int(os.environ.get("T", 7050)), 20004, 512
g = torch.Generator().manual_seed(1)
hid = (torch.randn(T, H, generator=g) * 0.5).to(torch.bfloat16).to(dev)
W = (torch.randn(V, H, generator=g) * 0.05).to(torch.bfloat16).to(dev)
b = (torch.randn(V, generator=g) * 0.1).to(dev)
t32 = torch.randint(1, V, (T,), generator=g).to(torch.int32).to(dev)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
ws = torch.zeros(lib.pv_vocab_fused_fwd_ws_bytes(T, V), dtype=torch.uint8, device=dev)
lse = torch.empty(T, device=dev); nll = torch.empty(T, device=dev); U = torch.empty(T, H, device=dev)
gs = (torch.rand(T, generator=g) + 0.5).to(dev)
ws2 = torch.zeros(lib.pv_vocab_fused_dw_ws_bytes(V), dtype=torch.uint8, device=dev)
dW = torch.zeros(V, H, device=dev); db = torch.zeros(V, device=dev)


def fwd():
    _lib.call("pv_vocab_fused_fwd", hid.data_ptr(), H, W.data_ptr(), H, b.data_ptr(), t32.data_ptr(), T, V, H, 1.0, -1,
              ws.data_ptr(), lse.data_ptr(), nll.data_ptr(), U.data_ptr(), None, st)


def dw():
    _lib.call("pv_vocab_fused_dw", hid.data_ptr(), H, W.data_ptr(), H, b.data_ptr(), t32.data_ptr(), lse.data_ptr(),
              gs.data_ptr(), T, V, H, 1.0, ws2.data_ptr(), dW.data_ptr(), H, 0, db.data_ptr(), None, st)


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1000 / n


print("T = %d live rows; fwd %.1f us, dw %.1f us per call (pair of launches each)" % (T, timed(fwd), timed(dw)))
if not hasattr(lib, "pv_vocab_stamps_copy"):
    sys.exit(0)
names = ["prologue (stationary loads, first tile)", "barrier (tile landed)", "first products (2 x 32 steps + staging)",
         "between the products (no MFMA in flight; 0 since the decision moved beside the second product)",
         "second products (2 x 32 steps) + decision + transform + staging", "segment tail (last sub-tile's second product)",
         "epilogue (slab stores)", "-"]
buf = np.zeros((2, 256, 4, 8), dtype=np.uint64)
lib.pv_vocab_stamps_copy.restype = C.c_int
lib.pv_vocab_stamps_copy.argtypes = [C.c_void_p, C.c_size_t]
fwd(); dw()
assert lib.pv_vocab_stamps_copy(buf.ctypes.data, buf.nbytes) == 0
for k, kn in enumerate(("vocab_fwd_dh_kernel", "vocab_dw_kernel")):
    x = buf[k].astype(np.float64)            # [wg][wave][section]
    tot = x.sum(-1)
    items = (T + 127) // 128 * ((V + 63) // 64) / 256 if k == 0 else (V + 127) // 128 * ((T + 63) // 64) / 256
    print("%s: cycles per wave: median %.0f, min %.0f, max %.0f; %.1f items per workgroup -> %.0f cycles per item" %
          (kn, np.median(tot), tot.min(), tot.max(), items, np.median(tot) / items))
    for i, n in enumerate(names[:7]):
        sec = x[:, :, i]
        print("   %-40s %5.1f %%   per item %6.0f   (per-wave spread %.0f .. %.0f)" %
              (n, 100 * sec.sum() / tot.sum(), np.median(sec) / items, sec.min() / items, sec.max() / items))
    for wv in range(4):
        print("   wave %d share of barrier wait: %.1f %%" % (wv, 100 * x[:, wv, 1].sum() / x[:, wv, :].sum()))
