"""Single-rank RCCL rehearsal of the collectives dp.FlatGradReducer issues (bucketed all_reduce on slices of a flat
buffer, all_gather_into_tensor of (ids, rows), broadcast), on the stream layout bench.py uses.  One GPU is enough: it
checks that RCCL initialises in this image and that the calls are well-formed; the exchange logic itself is covered by
tests/test_dp_gloo.py at world size 2.  Usage: python tools/rccl_selftest.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
from pevae_amd import dp

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
dev = torch.device("cuda", 0)
flat = torch.randn(5_000_000, device=dev)
ref = flat.clone()
red = dp.FlatGradReducer(flat, 1, sparse_tables=[(1000, 50, 8)], average=False, early_ranges=[(2_000_000, 3_000_000)])
works = red._all_reduce(red.early_bounds) + red._all_reduce(red.bounds)
ids = torch.randint(0, 50, (16,), device=dev)
pids, prows = dp.pack_sparse_rows(red.table_view(0), ids)
ids_all, rows_all = torch.empty_like(pids), torch.empty_like(prows)
works.append(dist.all_gather_into_tensor(ids_all, pids.contiguous(), async_op=True))
works.append(dist.all_gather_into_tensor(rows_all, prows.contiguous(), async_op=True))
for w in works:
    w.wait()
dp.merge_sparse_rows(red.table_view(0), pids, ids_all, rows_all)
dist.broadcast(flat, src=0)
torch.cuda.synchronize()
keep = torch.ones_like(flat, dtype=torch.bool)
keep[1000:1400] = False
assert torch.equal(flat[keep], ref[keep]), "single-rank sum must leave the dense stretches unchanged"
touched = torch.zeros(50, dtype=torch.bool, device=dev)
touched[ids] = True
assert torch.allclose(red.table_view(0)[touched], ref[1000:1400].view(50, 8)[touched])
# the reducer's own three-stage walk (start / start_mid on its own stream / finish), as bench.py drives it at N > 1
flat2 = torch.randn(3_000_000, device=dev)
flat2[1000:1400] = 0
ids2 = torch.randint(0, 50, (16,), device=dev)
flat2[1000:1400].view(50, 8)[ids2] = torch.randn(16, 8, device=dev)
ref2 = flat2.clone()
red2 = dp.FlatGradReducer(flat2, 1, sparse_tables=[(1000, 50, 8)], average=False, early_ranges=[(1_000_000, 2_000_000)],
                          mid_ranges=[(0, 1000), (1400, 500_000), (2_500_000, 3_000_000)], rehearse=True)
red2.start()
red2.start_mid([ids2])
red2.finish()
torch.cuda.synchronize()
assert torch.equal(flat2[keep[:3_000_000]], ref2[keep[:3_000_000]])
assert torch.allclose(flat2[1000:1400], ref2[1000:1400]), "one rank: the merged table equals the local one"
dist.destroy_process_group()
print("rccl selftest ok: %d early + %d dense buckets, sparse (ids, rows) gather, broadcast; three-stage walk %d / %d / %d buckets"
      % (len(red.early_bounds), len(red.bounds), len(red2.early_bounds), len(red2.mid_bounds), len(red2.bounds)))
