import sys; sys.path.insert(0, "/root/repo")
import torch
from pevae_amd import ops, _lib
dev = torch.device("cuda:0")
T, V, H = 12800, 20004, 512
gen = torch.Generator(device=dev).manual_seed(0)
hid = (torch.randn((T, H), device=dev, generator=gen) * 0.5).to(torch.bfloat16)
W = (torch.randn((V, H), device=dev, generator=gen) * 0.05).to(torch.bfloat16)
bias = torch.zeros(V, device=dev)
tgt = torch.randint(1, V, (T,), device=dev, generator=gen)
lib = _lib.load()
ws = torch.empty(lib.pv_vocab_nll_fwd_ws_bytes(T, V), dtype=torch.uint8, device=dev)
lse, nll = torch.empty(T, device=dev), torch.empty(T, device=dev)
p, st = ops._p, ops._stream
for live in (None, 12800, 6400, 1000):
    cnt = None if live is None else torch.tensor([live], dtype=torch.int32, device=dev)
    f = lambda: _lib.call("pv_vocab_nll_fwd", p(hid), H, p(W), H, p(bias), p(tgt), T, V, H, 1.0, 0, p(ws), p(lse), p(nll), p(cnt), st())
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print(live, "%.1f us" % (a.elapsed_time(b) * 100))
