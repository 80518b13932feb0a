"""One rank of tests/test_gpu_dls.py::test_global_negatives_two_ranks (not collected by pytest).  Both ranks share the one
GPU of the test box, so the group is gloo and dp.GatherRows stages its exchange through host memory.
argv: rank world port out_dir"""
import os
import sys
from argparse import Namespace

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build(dev, global_negatives):
    from pevae_amd import dls_latent as L
    cfg = Namespace(expand_weight=0.5, converge_weight=0.1, cip_weight=1.0, ost=False, no_mutual_information=False,
                    global_negatives=global_negatives)
    torch.manual_seed(11)
    dls = L.DLSAuto(4, 16, cfg).to(dev).train()
    den = L.DeNAuto(8, 16, cfg).to(dev).train()
    return dls, den


def inputs(rows, dev):
    gen = torch.Generator().manual_seed(12)
    mk = lambda *s: torch.randn(*s, generator=gen)
    d = dict(out=mk(rows, 4), rec=mk(rows, 4), a1=[mk(rows, k) for k in (32, 16, 8)], a2=[mk(rows, k) for k in (32, 16, 8)],
             noise=[torch.rand(rows, k, generator=gen) for k in (8, 16, 32)],
             d_out=mk(rows, 8), d_a1=[mk(rows, k) for k in (32, 16)], d_a2=[mk(rows, k) for k in (32, 16)],
             d_noise=[torch.rand(rows, k, generator=gen) for k in (16, 32)],
             gumbels=[-torch.empty(rows, rows).exponential_(generator=gen).log() for _ in range(2)])
    return d


def run(dls, den, d, lo, hi, dev):
    """Loss and flat parameter gradient for rows [lo, hi) of the inputs (the Gumbel draws stay global: [rows, rows])."""
    cut = lambda t: t[lo:hi].to(dev)
    z, rkl = dls(cut(d["out"]), cut(d["rec"]), [cut(t) for t in d["a1"]], [cut(t) for t in d["a2"]], False,
                 noise=[cut(t) for t in d["noise"]])
    z2, rkl2 = den(cut(d["d_out"]), [cut(t) for t in d["d_a1"]], [cut(t) for t in d["d_a2"]], False,
                   [cut(t) for t in d["d_noise"]], d["_gumbels"])
    loss = z.mean() * 3 + rkl + z2.mean() * 3 + rkl2
    loss.backward()
    ps = [p for m in (dls, den) for p in m.parameters()]
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
    return loss.detach().cpu(), flat.cpu()


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    dls, den = build(dev, True)
    B = 8
    d = inputs(world * B, dev)
    d["_gumbels"] = [g.to(dev) for g in d["gumbels"]]                      # the global [world*B, world*B] draw
    loss, flat = run(dls, den, d, rank * B, (rank + 1) * B, dev)
    dist.all_reduce(flat)
    dist.all_reduce(loss)
    torch.save((loss / world, flat / world), os.path.join(out, "gn%d.pt" % rank))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
