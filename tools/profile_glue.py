"""Which ATen glue launches are left in one eager training step (shapes and calling autograd node).
Run on the GPU box:  python tools/profile_glue.py > gpurun_out/glue.txt"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd.dls_vae import DLSVAE, default_config, synthetic_batch  # noqa: E402
from pevae_amd.optim import FlatAdam  # noqa: E402

dev = torch.device("cuda", 0)
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
cfg = default_config(dropout=0.3, batch_size=256, latent_stack="dls", use_attention=False)
torch.manual_seed(0)
model = DLSVAE(cfg).to(dev).train()
opt = FlatAdam(model.parameters(), lr=cfg.init_lr, max_norm=cfg.grad_clip, warmup_steps=cfg.warmup_steps)
feed = synthetic_batch(cfg, 256, 1234, dev)


def step():
    opt.zero_grad()
    loss = model(feed)
    model.valid_loss(loss).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA],
                            record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = None
count = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.device_time_total > 0 and not any(c.name.startswith("aten::") and c.device_time_total > 0 for c in e.cpu_children):
        stack = [s for s in (e.stack or []) if "/pevae_amd/" in s or "bench" in s][:3]
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        count[(e.name, str(e.input_shapes), par, " <- ".join(s.split("/pevae_amd/")[-1] for s in stack))] += 1
for (name, shapes, par, stack), n in sorted(count.items(), key=lambda kv: -kv[1]):
    print(n, name, shapes, "| parent:", par, "|", stack)
