"""Diagnostic: per-workgroup records of pv_greedy_decode at token 0 against a torch recomputation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden, load_params
from pevae_amd.enc2dec import decoders as D
from pevae_amd import ops
dev = torch.device("cuda:0")
g = load_golden("decoder_gen_greedy")
V, L, E, H, B = 37, 7, 12, 16, 4
emb = torch.nn.Embedding(V, E, padding_idx=0)
dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
dec.eval()
ids, h0 = torch.from_numpy(g["ids"]).to(dev), torch.from_numpy(g["h0"]).to(dev)
with torch.no_grad():
    logp, hid, ret = dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
    dec.persistent_decode = True
    lp, hn, ret_p = dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
torch.cuda.synchronize()
ws = ops.LAST_DECODE_WS[0]
Bpad, Hk, NV = 32, 32, 3
hb_bytes = (L + 1) * Bpad * Hk * 2
off = 4096 + (hb_bytes + 255) // 256 * 256
rec = ws[off:off + L * Bpad * NV * 16].cpu().numpy().view(np.float32).reshape(L, Bpad, NV, 4)
print("records t=0 row 0:", rec[0, 0], rec[0, 0].view(np.int32)[:, 3])
x = logp[0, 0].cpu().numpy()          # log-probs of row 0 token 0 (dense)
for r in range(NV):
    seg = x[r * 16:(r + 1) * 16]
    print("slab", r, "max", seg.max(), "sumexp(rel)", np.exp(seg - seg.max()).sum(), "argmax", r * 16 + seg.argmax())
print("lp persistent", lp[0], "dense", logp[0].max(-1)[0])
m, s_, b = rec[..., 0], rec[..., 1], rec[..., 2]
M = m.max(-1)
S = (s_ * np.exp(m - M[..., None])).sum(-1)
print("from records:", (b.max(-1) - M - np.log(S))[:, 0])
print("sum S:", S[:, 0], "M", M[:, 0])
print("kernel lp row0:", lp[0].cpu().numpy(), " -log(lp)->", np.exp(-lp[0].cpu().numpy()))
