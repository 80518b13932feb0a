"""What does one fork + join cost inside a replayed HIP graph?  Main chain: 60 sleep kernels (~8 us each).  At K points a
side branch with ONE tiny kernel is forked and joined again two main kernels later (the side work is long finished by
then), so any added time is the cross-queue hand-off itself."""
import sys
import time
import torch

dev = torch.device("cuda:0")
s, side = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1024, device=dev)
CYC, N = 20_000, int(sys.argv[2]) if len(sys.argv) > 2 else 60

JOIN = len(sys.argv) < 2 or sys.argv[1] != "forks-only"      # forks-only: the side branches are joined once, at the end
with torch.cuda.stream(s):
    base = None
    for K in ((0, 1, 5, 10, 20, 27) if N == 60 else (0, N // 8, N // 4, N // 2 - 4)):
        g = torch.cuda.CUDAGraph()
        x.add_(1)
        torch.cuda.synchronize()
        pts = set(range(2, N - 4, max(1, (N - 6) // K))[:K]) if K else set()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            pend = {}
            for i in range(N):
                torch.cuda._sleep(CYC)
                if i in pts:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        x.add_(1)
                    if JOIN:
                        pend[i + 2] = True
                if i in pend:
                    cur.wait_stream(side)
            cur.wait_stream(side)
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        ts, hs = [], []
        for _ in range(21):
            t0 = time.perf_counter()
            g.replay()
            hs.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts)[10] * 1e6
        print("   replay() returns to the host after %.0f us" % (sorted(hs)[10] * 1e6))
        base = t if base is None else base
        print("%2d fork/join pairs: graph takes %.0f us (+%.1f us per pair)" % (len(pts), t, (t - base) / max(1, len(pts))))
