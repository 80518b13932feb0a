"""Diagnostic: per-segment wall-clock shares of the persistent GRU backward step (needs the stamps build:
make -C pevae_amd/csrc stamps; PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so python tools/gru_bwd_stamps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import ops

dev = torch.device("cuda:0")
B, L, E, H = 256, 50, 100, 512
torch.manual_seed(0)
x = (torch.randn(B, L, E, device=dev) * 0.5).requires_grad_()
w_ih = (torch.randn(3 * H, E, device=dev) * 0.05).requires_grad_()
w_hh = (torch.randn(3 * H, H, device=dev) * 0.05).requires_grad_()
b = torch.zeros(3 * H, device=dev, requires_grad=True)
h0 = (torch.randn(B, H, device=dev) * 0.1).requires_grad_()
names = ["(a) loads issue", "poll", "barrier", "A loads+mfma", "part barrier", "combine+gates+barrier", "publish+drain",
         "barrier+arrive", "dgi stores issue", "(b) fragment loads issue"]
ops.TIMERS = {}
for _ in range(5):
    out, hn, _ = ops.GRULayer.apply(x, h0, w_ih, w_hh, b, b, True)
    ops.LAST_GRU_SYNC.clear()
    (out.sum() + hn.sum()).backward()
torch.cuda.synchronize()
print("timers:", {k: "%.1f us" % (sorted(a.elapsed_time(c) for a, c in v)[len(v) // 2] * 1e3) for k, v in ops.TIMERS.items()})
sync, NG, tag = ops.LAST_GRU_SYNC[-1]
print(tag)
words = sync.view(torch.int32)[64:64 + 160].cpu().view(16, 10)
for wg in range(16):
    print("  wg", wg, " ".join("%s=%.2f" % (n, words[wg, i].item() / 100.0 / L) for i, n in enumerate(names)),
          "| sum us/step %.2f" % (words[wg, :9].sum().item() / 100.0 / L))
