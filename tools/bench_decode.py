"""Free-running greedy decode (DecoderRNN mode=GEN) at the Yelp shape: time per decode of B sequences.
Usage: python tools/bench_decode.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pevae_amd.enc2dec import decoders as D

dev = torch.device("cuda:0")
V, L, E, H, B = 20004, 50, 100, 512, 256
torch.manual_seed(0)
emb = torch.nn.Embedding(V, E, padding_idx=0)
dec = D.DecoderRNN(V, L, E, H, 2, 3, 1, rnn_cell="gru", use_gpu=True, embedding=emb).to(dev)
dec.eval()
h0 = torch.randn(1, B, H, device=dev) * 0.1
ids = torch.zeros(B, L, dtype=torch.long, device=dev)
from pevae_amd import ops
for fused, graphed, persistent in ((False, False, False), (True, False, False), (True, True, False), (True, False, True)):
    dec.fused_greedy, dec.graph_decode, dec.persistent_decode = fused, graphed, persistent
    with torch.no_grad():
        for _ in range(2):
            dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            outs, hid, ret = dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ops.gru_check_all()
    print("fused_greedy=%s graph_decode=%s persistent_decode=%s: %.2f ms per decode of %d x %d tokens (%.1f us/token step), "
          "mean length %.1f" % (fused, graphed, persistent, dt * 1e3, B, L, dt * 1e6 / L, sum(ret["length"]) / B))
