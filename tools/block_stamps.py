"""Phase timing inside the fused Block kernels (diagnostic library from `make -C pevae_amd/csrc stamps`).
Usage: PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so python tools/block_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import dls_latent as L, _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    buf = (ctypes.c_ulonglong * 64)()
    names_f = ["start", "x+gelu|bar", "L0", "bar", "L1", "bar", "L2", "bar", "L3", "bar", "L4(end)", "-", "LN|bar"]
    for din, dout, mid in ((512, 256, 256), (256, 128, 128), (256, 512, 512), (64, 32, 32)):
        blk = L.Block(din, dout, mid).to(dev).train()
        x = torch.randn(256, din, device=dev)
        w = torch.randn(256, dout, device=dev)
        for it in range(3):
            xi = x.clone().requires_grad_()
            y = blk(xi)
            (y * w).sum().backward()
        torch.cuda.synchronize()
        lib.pv_block_stamps_read.argtypes = [ctypes.c_void_p]
        assert lib.pv_block_stamps_read(buf) == 0
        for k, nm in ((0, "fwd"), (1, "bwd")):
            t = [buf[k * 32 + i] for i in range(13)]
            order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 10] if k == 0 else [0, 1, 2, 3, 12, 4, 5, 6, 7, 8, 9, 10]
            if k == 0 and t[11] >= t[10]:          # the 4-row kernel: [even] = products done, [odd] = epilogue + barrier, [11] = end
                order.append(11)
            seq = [(i, t[i]) for i in order]
            print("Block(%d,%d,%d) %s: total %.2f us" % (din, dout, mid, nm, (seq[-1][1] - seq[0][1]) / 100.0))
            if k == 0:
                cyc = buf[15] - buf[14]
                print("   shader clock during the kernel: %.0f MHz" % (cyc / ((max(t[10], t[11]) - t[0]) / 100.0)))
            print("   " + "  ".join("[%d]+%.2f" % (seq[j][0], (seq[j][1] - seq[j - 1][1]) / 100.0) for j in range(1, len(seq))))


if __name__ == "__main__":
    main()
