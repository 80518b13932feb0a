"""Batch-sharded data parallelism for the DLS_VAE step: one process per GPU, gradients exchanged with bucketed
RCCL all-reduces over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

The reference is single-process / single-GPU (main_re.py:200), so this layer has no counterpart there.
Each rank runs the full model on its own 256 samples; the only exchange is the gradient reduction over the flat
gradient buffer of FlatAdam.  The 'seq'-averaged NLL divides by the LOCAL batch (criterions.py:64), so summed
gradients are divided by the world size; the clip norm / skip guards are then computed on the reduced gradient,
which is identical on every rank, so all ranks take the same decision without a further exchange.
xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): buckets are sized (default 16 MiB) so several are in
flight across the links while keeping launch overhead negligible.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def bucket_bounds(n, bucket_elems):
    """[(start, end)] covering [0, n) in order."""
    return [(s, min(s + bucket_elems, n)) for s in range(0, n, bucket_elems)]


class FlatGradReducer(object):
    """Averages a flat fp32 gradient buffer across ranks in fixed-size buckets (async, then one wait)."""

    def __init__(self, flat_grad, world_size, bucket_bytes=16 << 20, group=None):
        self.flat, self.world, self.group = flat_grad, world_size, group
        self.bounds = bucket_bounds(flat_grad.numel(), max(1, bucket_bytes // 4))

    def reduce(self):
        if self.world <= 1:
            return
        works = [dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for s, e in self.bounds]
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / self.world)


def broadcast_parameters(flat_params, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
