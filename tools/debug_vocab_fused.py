"""Diagnostic: csrc/vocab_fused.hip against a float64 reference, piece by piece (lse, target logit, U, dW, dbias)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import _lib, ops

dev = torch.device("cuda:0")
T, V, H = int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 2003, 512
g = torch.Generator().manual_seed(1)
hid = (torch.randn(T, H, generator=g) * 0.5).to(torch.bfloat16)
W = (torch.randn(V, H, generator=g) * 0.05).to(torch.bfloat16)
b = torch.randn(V, generator=g) * 0.1
tgt = torch.randint(1, V, (T,), generator=g)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
hd, Wd, bd = hid.to(dev), W.to(dev), b.to(dev)
t32 = tgt.to(torch.int32).to(dev)
ws = torch.zeros(lib.pv_vocab_fused_fwd_ws_bytes(T, V), dtype=torch.uint8, device=dev)
lse = torch.empty(T, device=dev); nll = torch.empty(T, device=dev); U = torch.empty(T, H, device=dev)
_lib.call("pv_vocab_fused_fwd", hd.data_ptr(), H, Wd.data_ptr(), H, bd.data_ptr(), t32.data_ptr(), T, V, H, 1.0, -1,
          ws.data_ptr(), lse.data_ptr(), nll.data_ptr(), U.data_ptr(), None, st)
torch.cuda.synchronize()
logits = hid.double() @ W.double().t() + b.double()
lse_r = torch.logsumexp(logits, 1)
tl_r = logits.gather(1, tgt[:, None])[:, 0]
P = torch.softmax(logits, 1)
U_r = P @ W.double() - W.double()[tgt]
print("lse  max err", float((lse.cpu().double() - lse_r).abs().max()), "first", lse[:4].tolist(), lse_r[:4].tolist())
print("nll  max err", float((nll.cpu().double() - (lse_r - tl_r)).abs().max()))
print("U    rel err", float((U.cpu().double() - U_r).abs().max() / U_r.abs().max()))
gs = torch.rand(T, generator=g) + 0.5
ws2 = torch.zeros(lib.pv_vocab_fused_dw_ws_bytes(V), dtype=torch.uint8, device=dev)
dW = torch.zeros(V, H, device=dev); db = torch.zeros(V, device=dev)
lse_d, gs_d = lse_r.float().to(dev), gs.to(dev)
_lib.call("pv_vocab_fused_dw", hd.data_ptr(), H, Wd.data_ptr(), H, bd.data_ptr(), t32.data_ptr(), lse_d.data_ptr(),
          gs_d.data_ptr(), T, V, H, 1.0, ws2.data_ptr(), dW.data_ptr(), H, 0, db.data_ptr(), None, st)
torch.cuda.synchronize()
Pg = (P - torch.nn.functional.one_hot(tgt, V).double()) * gs.double()[:, None]
dW_r = Pg.t() @ hid.double()
db_r = Pg.sum(0)
print("dW   rel err", float((dW.cpu().double() - dW_r).abs().max() / dW_r.abs().max()))
print("db   rel err", float((db.cpu().double() - db_r).abs().max() / db_r.abs().max()))
# timing
import time
for name, fn in (("fwd", lambda: _lib.call("pv_vocab_fused_fwd", hd.data_ptr(), H, Wd.data_ptr(), H, bd.data_ptr(), t32.data_ptr(), T, V, H, 1.0, -1,
          ws.data_ptr(), lse.data_ptr(), nll.data_ptr(), U.data_ptr(), None, st)),
                 ("dw", lambda: _lib.call("pv_vocab_fused_dw", hd.data_ptr(), H, Wd.data_ptr(), H, bd.data_ptr(), t32.data_ptr(), lse_d.data_ptr(),
          gs_d.data_ptr(), T, V, H, 1.0, ws2.data_ptr(), dW.data_ptr(), H, 0, db.data_ptr(), None, st))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s: %.1f us  -> %.0f TFLOP/s on 2 products" % (name, ms * 1e3, 4.0 * T * V * H / (ms * 1e-3) / 1e12))
if len(sys.argv) > 3:
    e = (dW.cpu().double() - dW_r).abs()
    print("dW err by row block of 32:", [round(float(e[i:i + 32].max()), 4) for i in range(0, min(V, 256), 32)])
    print("dW err by col block of 32:", [round(float(e[:, i:i + 32].max()), 4) for i in range(0, H, 32)])
    eu = (U.cpu().double() - U_r).abs()
    print("U err by row block of 32:", [round(float(eu[i:i + 32].max()), 4) for i in range(0, min(T, 256), 32)])
    print("U err by col block of 32:", [round(float(eu[:, i:i + 32].max()), 4) for i in range(0, H, 32)])
    el = (lse.cpu().double() - lse_r).abs()
    print("lse err by row block:", [round(float(el[i:i + 32].max()), 4) for i in range(0, min(T, 256), 32)])
