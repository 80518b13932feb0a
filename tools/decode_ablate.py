"""Diagnostic: time of one persistent greedy decode with parts of the kernel switched off (PV_DECODE_DBG bits: 1 vocab A
loads, 2 vocab MFMAs, 4 epilogue math, 8 record stores, 16 reduce loads, 32 GRU MFMAs, 64 all waits; results are wrong
with any bit set; the bits exist only in the ablation build: make -C pevae_amd/csrc ablate).
Usage: PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_ablate.so PV_DECODE_DBG=<bits> python tools/decode_ablate.py"""
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
for _ in range(3):
    ops.greedy_decode(emb, w_ih, w_hh, b, b, wp, bp, h0, L, 2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.greedy_decode(emb, w_ih, w_hh, b, b, wp, bp, h0, L, 2)
e1.record()
torch.cuda.synchronize()
print("PV_DECODE_DBG=%s: %.1f us per decode call, %.2f us per token" % (os.environ.get("PV_DECODE_DBG", "0"),
      e0.elapsed_time(e1) * 100, e0.elapsed_time(e1) * 100 / L))
