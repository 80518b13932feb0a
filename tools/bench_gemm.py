"""Micro-benchmark of the bf16 MFMA GEMM core on random operands: square sizes (to compare with the known
ladder of this structure) and the four vocabulary-projection shapes of the training step.
Usage: python tools/bench_gemm.py [reps]"""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import ops, _lib  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)

    def rnd(m, k):
        return (torch.rand((m, k), device=dev, generator=gen) * 2 - 1).to(BF16)

    print("%-34s %10s %10s %10s" % ("shape (M,N,K)", "med us", "min us", "TF (med)"))
    shapes = [("square", 4096, 4096, 4096), ("square", 8192, 8192, 8192),
              ("vocab fwd-like (plain store bf16)", 12800, 20032, 512),
              ("dH: P W^T", 12800, 512, 20032), ("dW: P^T H^T", 20004, 512, 12800)]
    for name, M, N, K in shapes:
        A, B = rnd(M, K), rnd(N, K)
        out = torch.empty((M, N), dtype=BF16, device=dev)
        med, mn = timeit(lambda: ops.gemm_nt(A, K, B, K, M, N, K, out_bf16=out), reps)
        print("%-34s %10.1f %10.1f %10.1f   %s" % ("%d,%d,%d" % (M, N, K), med, mn, 2.0 * M * N * K / med / 1e6, name))
        if name.startswith("d"):
            o32 = torch.zeros((M, N), dtype=torch.float32, device=dev)
            med, mn = timeit(lambda: ops.gemm_nt_wgrad(A, K, B, K, M, N, K, sink=None), reps)
            print("%-34s %10.1f %10.1f %10.1f   %s" % ("%d,%d,%d" % (M, N, K), med, mn, 2.0 * M * N * K / med / 1e6, name + " split-K"))
        del A, B, out

    # the fused vocabulary kernels at the step's shape
    T, V, H = 12800, 20004, 512
    hid = (torch.randn((T, H), device=dev, generator=gen) * 0.5).to(BF16)
    W = (torch.randn((V, H), device=dev, generator=gen) * 0.05).to(BF16)
    bias = torch.zeros(V, device=dev)
    tgt = torch.randint(1, V, (T,), device=dev, generator=gen)
    lib = _lib.load()
    ws = torch.empty(lib.pv_vocab_nll_fwd_ws_bytes(T, V), dtype=torch.uint8, device=dev)
    lse, nll = torch.empty(T, device=dev), torch.empty(T, device=dev)
    p, st = ops._p, ops._stream
    med, mn = timeit(lambda: _lib.call("pv_vocab_nll_fwd", p(hid), H, p(W), H, p(bias), p(tgt), T, V, H, 1.0, 0, p(ws), p(lse),
                                       p(nll), None, st()), reps)
    print("%-34s %10.1f %10.1f %10.1f   %s" % ("vocab_nll_fwd", med, mn, 2.0 * T * V * H / med / 1e6, "fused softmax partials"))
    Tp, Vp = ops.round_up(T, 64), ops.round_up(V, 64)
    P = torch.empty((Tp, Vp), dtype=BF16, device=dev)
    PT = torch.empty((Vp, Tp), dtype=BF16, device=dev)
    dbp = torch.empty((Tp // 64, Vp), device=dev)
    dbias = torch.empty(V, device=dev)
    gs = torch.full((T,), 1.0 / 256, device=dev)
    med, mn = timeit(lambda: _lib.call("pv_vocab_nll_bwd_prob", p(hid), H, p(W), H, p(bias), p(tgt), p(lse), p(gs), T, V, H, 1.0,
                                       p(P), Vp, p(PT), Tp, p(dbp), p(dbias), None, st()), reps)
    print("%-34s %10.1f %10.1f %10.1f   %s" % ("vocab_nll_bwd_prob", med, mn, 2.0 * T * V * H / med / 1e6, "P, P^T, dbias"))


if __name__ == "__main__":
    main()
