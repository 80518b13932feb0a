"""Timing ablation of the persistent GRU forward kernel (PV_GRU_DBG bits; results are wrong by design)."""
import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from pevae_amd import ops
    dev = torch.device("cuda:0")
    B, L, E, H = 256, 50, 100, 512
    torch.manual_seed(0)
    x = torch.randn(B, L, E, device=dev) * 0.5
    w_ih = torch.randn(3 * H, E, device=dev) * 0.05
    w_hh = torch.randn(3 * H, H, device=dev) * 0.05
    b = torch.zeros(3 * H, device=dev)
    h0 = torch.randn(B, H, device=dev) * 0.1
    for save in (True, False):
        ops.TIMERS = {}
        for _ in range(30):
            with torch.no_grad():
                ops.GRULayer.apply(x, h0, w_ih, w_hh, b, b, save)
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(c) for a, c in ops.TIMERS["gru_fwd"][5:])
        if os.environ.get("PV_GRU_SERIES"):
            print("series", " ".join("%.0f" % (a.elapsed_time(c) * 1e3) for a, c in ops.TIMERS["gru_fwd"]))
        print("dbg=%s save=%d median %.1f us  min %.1f us" % (sys.argv[1], save, ts[len(ts) // 2] * 1e3, ts[0] * 1e3), flush=True)
else:
    for d in ([int(v) for v in os.environ['PV_GRU_ONLY'].split(',')] if os.environ.get('PV_GRU_ONLY') else (0, 1, 1 + 2, 1 + 2 + 4, 1 + 2 + 4 + 8 + 16, 1 + 2 + 4 + 8 + 16 + 32, 127, 32, 2, 8, 4, 64, 1 + 16)):
        subprocess.run([sys.executable, __file__, str(d)], env=dict(os.environ, PV_GRU_DBG=str(d)))
