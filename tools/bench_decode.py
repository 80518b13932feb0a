"""Free-running greedy / sample / beam decodes (DecoderRNN mode=GEN) at the Yelp shape: time per decode of B sequences.
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
cases = [("greedy", 1, False, False, False), ("greedy", 1, True, False, False), ("greedy", 1, True, True, False),
         ("greedy", 1, True, False, True), ("sample", 1, False, False, False), ("sample", 1, True, False, False),
         ("sample", 1, True, True, False), ("sample", 1, True, False, True), ("beam", 4, False, False, False),
         ("beam", 4, True, False, False), ("beam", 4, True, True, False)]
for gen_type, beam, fused, graphed, persistent in cases:
    dec.fused_greedy, dec.graph_decode, dec.persistent_decode = fused, graphed, persistent
    Bq = B // beam                                  # beam: the same B x L token steps in all
    with torch.no_grad():
        for _ in range(2):
            dec(Bq, ids[:Bq], h0[:, :Bq].contiguous(), mode=D.GEN, gen_type=gen_type, beam_size=beam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            outs, hid, ret = dec(Bq, ids[:Bq], h0[:, :Bq].contiguous(), mode=D.GEN, gen_type=gen_type, beam_size=beam)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ops.gru_check_all()
    print("%-6s fused=%s graph_decode=%s persistent_decode=%s: %.2f ms per decode of %d x %d token steps (%.1f us/token step)"
          % (gen_type, fused, graphed, persistent, dt * 1e3, Bq * beam, L, dt * 1e6 / L))
