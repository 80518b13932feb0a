"""Diagnostic: per-segment wall-clock shares of the persistent GRU forward step (needs the stamps build:
make -C pevae_amd/csrc stamps; PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so python tools/gru_stamps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import ops

dev = torch.device("cuda:0")
B, L, E, H = 256, 50, 100, 512
torch.manual_seed(0)
x = torch.randn(B, L, E, device=dev) * 0.5
w_ih = torch.randn(3 * H, E, device=dev) * 0.05
w_hh = torch.randn(3 * H, H, device=dev) * 0.05
b = torch.zeros(3 * H, device=dev)
h0 = torch.randn(B, H, device=dev) * 0.1
names = ["prefetch issue", "poll", "barrier", "A loads landed", "mfma", "gates+tile+barrier", "publish+drain",
         "barrier+arrive", "deferred stores issue", "gates+tile (before the barrier)"]
for save in (True, False):
    for _ in range(5):
        with torch.no_grad():
            ops.GRULayer.apply(x, h0, w_ih, w_hh, b, b, save)
    torch.cuda.synchronize()
    sync, NG, _ = ops.LAST_GRU_SYNC[-1]
    words = sync.view(torch.int32)[64:64 + 160].cpu().view(16, 10)
    print("save =", save)
    for wg in range(16):
        print("  wg", wg, " ".join("%s=%.2f" % (n, words[wg, i].item() / 100.0 / L) for i, n in enumerate(names)),
              "| sum us/step %.2f" % (words[wg, :10].sum().item() / 100.0 / L))
    ops.LAST_GRU_SYNC.clear()
