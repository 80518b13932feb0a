"""Host cost of hipGraphLaunch as a function of the node count: graphs of N tiny kernels, replay() timed with the GPU
idle (synchronised before each call) and back to back."""
import time
import torch

dev = torch.device("cuda:0")
s = torch.cuda.Stream()
x = torch.zeros(1024, device=dev)
with torch.cuda.stream(s):
    for n in (25, 75, 170, 340):
        g = torch.cuda.CUDAGraph()
        x.add_(1)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                x.add_(1)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        idle = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            idle.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%3d nodes: replay() on an idle GPU %.0f us (%.2f us/node); 50 back to back: host %.0f us each, GPU-complete %.0f us each"
              % (n, sorted(idle)[10] * 1e6, sorted(idle)[10] * 1e6 / n, (t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))

# does replay() block the host?  a chain of 40 sleep kernels (~0.1 ms each), single stream vs with a forked side branch
side = torch.cuda.Stream()
CYC = 200_000
with torch.cuda.stream(s):
    for fork in (0, 1, 4):
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            cur = torch.cuda.current_stream()
            for i in range(40):
                torch.cuda._sleep(CYC)
                if fork and i % (40 // fork) == 0:
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(CYC)
                        torch.cuda._sleep(CYC)
                    if i % 2 == 0:
                        cur.wait_stream(side)
            cur.wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("40-kernel chain, %d forks: replay() returns after %.0f us; the graph takes %.0f us on the GPU"
              % (fork, (t1 - t0) * 1e6, (t2 - t0) * 1e6))
