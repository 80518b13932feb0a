"""Micro-benchmark of the persistent GRU kernels alone (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd import ops

dev = torch.device("cuda:0")
B, L, E, H = 256, 50, 100, 512
torch.manual_seed(0)
x = torch.randn(B, L, E, device=dev) * 0.5
w_ih = (torch.randn(3 * H, E, device=dev) * 0.05).requires_grad_()
w_hh = (torch.randn(3 * H, H, device=dev) * 0.05).requires_grad_()
b_ih = torch.zeros(3 * H, device=dev, requires_grad=True)
b_hh = torch.zeros(3 * H, device=dev, requires_grad=True)
h0 = torch.randn(B, H, device=dev) * 0.1


def run(save, n=20):
    ops.TIMERS = {}
    for _ in range(n):
        if save:
            out, hn, _ = ops.GRULayer.apply(x, h0, w_ih, w_hh, b_ih, b_hh, True)
            (out.sum() + hn.sum()).backward()
        else:
            with torch.no_grad():
                ops.GRULayer.apply(x, h0, w_ih, w_hh, b_ih, b_hh, False)
    torch.cuda.synchronize()
    ops.gru_check_all()
    res = {}
    for tag, evs in ops.TIMERS.items():
        ts = sorted(a.elapsed_time(b) for a, b in evs[3:])
        res[tag] = (ts[len(ts) // 2] * 1e3, ts[0] * 1e3)
    ops.TIMERS = None
    return res


for wgs in (0, 64, 32):
    ops.GRU_MAX_WGS = wgs
    print("max_wgs", wgs, "train:", run(True), "infer:", run(False))
