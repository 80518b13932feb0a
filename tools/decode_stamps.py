"""Diagnostic: per-segment wall-clock shares of a token of the persistent greedy decode (needs the stamps build:
make -C pevae_amd/csrc stamps; PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so python tools/decode_stamps.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import ops

dev = torch.device("cuda:0")
V, L, E, H, B = 20004, 50, 100, 512, 256
torch.manual_seed(0)
emb = torch.randn(V, E, device=dev) * 0.1
w_ih, w_hh = torch.randn(3 * H, E, device=dev) * 0.04, torch.randn(3 * H, H, device=dev) * 0.04
b = torch.zeros(3 * H, device=dev)
wp, bp = torch.randn(V, H, device=dev) * 0.1, torch.zeros(V, device=dev)
h0 = torch.randn(B, H, device=dev) * 0.1
names = ["wait B", "reduce", "gru step", "publish h", "wait A", "vocab mfma", "epilogue", "publish records"]
for _ in range(3):
    ops.greedy_decode(emb, w_ih, w_hh, b, b, wp, bp, h0, L, 2)
torch.cuda.synchronize()
ops.gru_check_all.__globals__["LAST_GRU_SYNC"].clear()
ws = ops.LAST_DECODE_WS[0]
words = ws[:4096].view(torch.int32)[576:576 + 64].cpu().view(8, 8)
for k in range(8):
    print("wg %s" % ("%d" % k if k < 4 else "last-%d" % (7 - k)),
          " ".join("%s=%.2f" % (n, words[k, i].item() / 100.0 / L) for i, n in enumerate(names)),
          "| sum us/token %.2f" % (words[k, :8].sum().item() / 100.0 / L))
