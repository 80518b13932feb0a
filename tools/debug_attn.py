import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_modules as R
from pevae_amd.enc2dec.decoders import Attention
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Lo, Li, D) in [(24, 11, 11, 64), (4, 7, 5, 16), (8, 50, 50, 512)]:
    a = Attention(D, D, "cat", project=True)
    r = R.Attention(D, D, "cat", project=True)
    r.load_state_dict(a.state_dict())
    a = a.to(dev)
    out, ctx = torch.randn(B, Lo, D) * 0.5, torch.randn(B, Li, D) * 0.5
    w = torch.randn(B, Lo, D)
    o1, c1 = out.clone().requires_grad_(), ctx.clone().requires_grad_()
    y, at = r(o1, c1); (y * w).sum().backward()
    o2, c2 = out.to(dev).requires_grad_(), ctx.to(dev).requires_grad_()
    y2, at2 = a(o2, c2); (y2 * w.to(dev)).sum().backward()
    print(B, Lo, Li, D, "y", float((y2.cpu() - y).abs().max()), "attn", float((at2.cpu() - at).abs().max()))
    for (n, p), (_, q) in zip(a.named_parameters(), r.named_parameters()):
        g, h = p.grad.cpu(), q.grad
        print("   ", n, "maxerr %.3e scale %.3e cos %.6f" % (float((g - h).abs().max()), float(h.abs().max()),
              float((g * h).sum() / (g.norm() * h.norm() + 1e-20))))
    print("    dout", float((o2.grad.cpu() - o1.grad).abs().max()), "dctx", float((c2.grad.cpu() - c1.grad).abs().max()))
