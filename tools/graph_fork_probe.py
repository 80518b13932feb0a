"""How does the HIP-graph executor run a fork?  A0 -> {A1 A2 A3 | B1 B2 B3} -> join, every node a ~100 us spin kernel.
Ideal replay: ~400 us; serialised: ~700 us.  Variants differ only in CAPTURE ORDER of the two branches."""
import sys
import torch

dev = torch.device("cuda:0")
CYC = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000     # torch.cuda._sleep cycles (100 MHz counter on AMD? measured below)


def spin():
    torch.cuda._sleep(CYC)


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


main = torch.cuda.Stream()
side = torch.cuda.Stream()
with torch.cuda.stream(main):
    one = time_fn(spin)
    print("one spin kernel: %.1f us" % one)

    def build(order):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            cur = torch.cuda.current_stream()
            spin()                                   # A0
            ev = torch.cuda.Event()
            ev.record(cur)
            if order == "side_first":
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    spin(); spin(); spin()
                spin(); spin(); spin()
            elif order == "main_first":
                spin(); spin(); spin()
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    spin(); spin(); spin()
            elif order == "interleaved":
                side.wait_event(ev)
                for _ in range(3):
                    spin()
                    with torch.cuda.stream(side):
                        spin()
            elif order == "serial":
                for _ in range(6):
                    spin()
            cur.wait_stream(side)
            spin()                                   # tail
        return g

    for order in ("serial", "side_first", "main_first", "interleaved"):
        g = build(order)
        t = time_fn(g.replay)
        print("%-12s replay %.1f us  (= %.2f kernels)" % (order, t, t / one))
