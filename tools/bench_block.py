"""Time one Block forward+backward, fused (pv_block_fwd/bwd) vs per-linear launches.  python tools/bench_block.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import dls_latent as L, ops  # noqa: E402


def run(blk, x, w, n):
    for p in blk.parameters():
        p.grad = torch.zeros_like(p)
    def step():
        xi = x.clone().requires_grad_()
        y = blk(xi)
        (y * w).sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        step()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    dev = torch.device("cuda:0")
    for din, dout, mid in ((512, 256, 256), (256, 128, 128), (128, 64, 64), (64, 32, 32), (96, 128, 128), (256, 512, 512)):
        blk = L.Block(din, dout, mid).to(dev).train()
        x = torch.randn(256, din, device=dev)
        w = torch.randn(256, dout, device=dev)
        res = []
        for fused in (True, False):
            ops.FUSED_BLOCK = fused
            res.append(run(blk, x, w, 30))
        ops.FUSED_BLOCK = True
        print("Block(%d,%d,%d) M=256  fused %.1f us   unfused %.1f us  (eager, incl. launch overhead)" % (din, dout, mid, res[0], res[1]))


if __name__ == "__main__":
    main()
