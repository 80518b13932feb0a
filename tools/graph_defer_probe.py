"""When does the HIP-graph executor START a side branch whose launches were captured late?
A0 -> main chain of N short kernels -> join -> tail;  side branch of M long kernels depends on A0 only.
Variants: where in CAPTURE order the side branch sits (at once / after k main kernels / after all of them).
Ideal replay = A0 + max(N*short, M*long) + tail."""
import sys
import torch

dev = torch.device("cuda:0")
SHORT, LONG, N, M = 2_000_000, 6_000_000, 30, 8     # torch.cuda._sleep cycles


def time_fn(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


main, side = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    ts = time_fn(lambda: torch.cuda._sleep(SHORT))
    tl = time_fn(lambda: torch.cuda._sleep(LONG))
    print("short %.1f us, long %.1f us; main chain %d short = %.0f us, side %d long = %.0f us" % (ts, tl, N, N * ts, M, M * tl))

    def build(at):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            cur = torch.cuda.current_stream()
            torch.cuda._sleep(SHORT)                     # A0
            ev = torch.cuda.Event()
            ev.record(cur)
            for i in range(N + 1):
                if i == at:
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        for _ in range(M):
                            torch.cuda._sleep(LONG)
                if i < N:
                    torch.cuda._sleep(SHORT)
            cur.wait_stream(side)
            torch.cuda._sleep(SHORT)                     # tail
        return g

    for at in (0, 1, N // 2, N):
        g = build(at)
        t = time_fn(g.replay)
        print("side branch captured after %2d main kernels: replay %.1f us (ideal %.0f, serial %.0f)" % (
            at, t, 2 * ts + max(N * ts, M * tl), (N + 2) * ts + M * tl))
