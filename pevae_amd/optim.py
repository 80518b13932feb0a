"""Optimizer tail of the training step on device: global-norm clip, the reference's skip guards and Adam over
ONE flat buffer (main_pretrain.py:207-225,268-269; dialog_models.py:122-131), two kernel launches + one small
guard kernel per step and no host synchronisation."""
import os
import struct

import torch

from . import _lib
from .ops import _p, _stream

ZERO_BESIDE = os.environ.get("PEVAE_ZERO_BESIDE", "1") == "1"      # A/B switch of zero_grad(beside_forward=True)


class SkipFlag(object):
    def __init__(self, buf, ev):
        self.buf, self.ev, self._val = buf, ev, None

    def ready(self):
        return self._val is not None or self.ev.query()

    def _read(self):
        if self._val is None:
            self.ev.synchronize()
            self._val = (int(self.buf[0]), struct.unpack("f", struct.pack("i", int(self.buf[1])))[0])
        return self._val

    def skipped(self):
        return self._read()[0] != 0

    def lr(self):
        return self._read()[1]


class FlatAdam(object):
    """Re-homes every parameter (and its .grad) as a view into flat fp32 buffers, then steps them with
    pv_grad_guard + pv_adam_step.  torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay.

    lr schedule: transformers.get_constant_schedule_with_warmup -- the k-th applied step (k = 0, 1, ...) runs
    at lr * min(1, k / warmup_steps).  `guards=True` adds the reference's skip rules (non-finite norm, up to 5
    in a row; norm > 3x its running mean).  `weight_decay` > 0 makes it torch.optim.AdamW (decoupled decay; the
    reference's freeze-phase optimizer, dialog_models.py:834, uses AdamW's default 0.01).  `grad_scale`: the flat
    gradient holds a sum over `1 / grad_scale` data-parallel ranks (dp.FlatGradReducer); the scale is applied inside
    the guard / Adam kernels."""

    def __init__(self, params, lr, max_norm, warmup_steps=0, betas=(0.9, 0.999), eps=1e-8, guards=True,
                 weight_decay=0.0, grad_scale=1.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam needs parameters on an MI355X device; there is no CPU fallback")
        self.max_norm, self.warmup, self.betas, self.eps, self.guards = max_norm, warmup_steps, betas, eps, guards
        self.weight_decay, self.grad_scale = float(weight_decay), float(grad_scale)
        self._flag_ring, self._flag_slot = [], 0
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4            # keep every view 16-byte aligned
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + p.numel()].view(p.shape)
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.offsets = offs
        self.state = torch.zeros(_lib.load().pv_optim_state_bytes(), dtype=torch.uint8, device=dev)
        self.lr = lr

    @property
    def lr(self):
        """The base learning rate.  It lives in device memory (state block, byte 48) and the guard kernel reads it from
        there, so a step captured in a HIP graph follows later assignments (engine.adjust_learning_rate at a plateau)."""
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self.state[48:52].view(torch.float32).fill_(self._lr)          # stream-ordered with the steps around it

    def table_range(self, param):
        """(offset, rows, dim) of a 2-D parameter inside the flat buffers (dp.FlatGradReducer sparse_tables)."""
        for p, o in zip(self.params, self.offsets):
            if p is param:
                return (o, p.shape[0], p.shape[1])
        raise KeyError("FlatAdam.table_range: not one of this optimizer's parameters")

    def param_range(self, param):
        """(start, end) of a parameter inside the flat buffers (dp.FlatGradReducer early_ranges)."""
        for p, o in zip(self.params, self.offsets):
            if p is param:
                return (o, o + p.numel())
        raise KeyError("FlatAdam.param_range: not one of this optimizer's parameters")

    def spans(self, params):
        """[(start, end)] covering the given parameters inside the flat buffers, alignment padding included, neighbours
        merged (dp.FlatGradReducer early_ranges / mid_ranges)."""
        want = set(id(p) for p in params)
        out = []
        for p, o in zip(self.params, self.offsets):
            if id(p) in want:
                e = o + (p.numel() + 3) // 4 * 4
                if out and out[-1][1] == o:
                    out[-1] = (out[-1][0], e)
                else:
                    out.append((o, e))
        return out

    def zero_grad(self, beside_forward=False):
        """beside_forward: the 124 MB fill runs on the ops side stream, forked here and joined by the first join_side_work()
        of the forward pass that MUST follow (every recurrence launch makes one) -- it then runs beside the launches at the
        head of the step instead of in front of them.  Only for callers that run a forward pass between this call and the
        backward pass (bench.py, engine.GraphedStep)."""
        if beside_forward and self.flat_g.is_cuda and ZERO_BESIDE:
            from . import ops
            with ops.fork_side():
                self.flat_g.zero_()
        else:
            self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-attach the view
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def guard(self):
        """Clip norm + the skip rules of main_pretrain.py:207-225 on the device (decision and coefficients into `state`)."""
        _lib.call("pv_grad_guard", _p(self.flat_g), self.n, _p(self.state), self.max_norm, -1.0, self.warmup,
                  self.betas[0], self.betas[1], 1 if self.guards else 0, self.grad_scale, _stream())

    def apply(self):
        """optimizer.step() + scheduler.step() (:268-269) unless the guard dropped this step."""
        _lib.call("pv_adam_step", _p(self.flat_p), _p(self.flat_g), _p(self.exp_avg), _p(self.exp_avg_sq), self.n,
                  _p(self.state), self.betas[0], self.betas[1], self.eps, self.weight_decay, _stream())

    def step(self):
        self.guard()
        self.apply()

    def skip_flag_async(self):
        """Non-blocking read-back of this step's skip decision (main_pretrain.py:213-222 does `continue` on it): an
        8-byte copy {skip, lr_eff} into one of 8 page-locked slots, ordered behind the step on the current stream.
        -> handle with .ready() (poll), .skipped() (waits for the copy) and .lr()."""
        if len(self._flag_ring) < 8:
            self._flag_ring.append([torch.empty(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event(), None])
        slot = self._flag_ring[self._flag_slot % len(self._flag_ring)]
        self._flag_slot += 1
        buf, ev, last = slot
        if last is not None:
            # The handle given out 8 steps ago may not have been read yet (a host that runs ahead of the device keeps
            # unread handles): it takes its value NOW -- waiting for its copy -- before the event is re-armed and the
            # buffer overwritten, so every handle reports its own step whenever it is read.
            last._read()
        buf.copy_(self.state[28:36].view(torch.int32), non_blocking=True)
        ev.record()
        slot[2] = flag = SkipFlag(buf, ev)
        return flag

    def state_dict_host(self):
        """Copy the device-side guard state to the host (synchronises; logging / tests only)."""
        raw = bytes(self.state[:48].cpu().numpy())
        avg, have, jump, cnt, t, norm, coef, skip, lr, bc1, bc2, skipped = struct.unpack("fiiiiffifffi", raw)
        return dict(avg_norm=avg, jump_cnt=jump, batch_cnt=cnt, adam_step=t, last_norm=norm, clip_coef=coef,
                    skip=skip, lr=lr, skipped_total=skipped)
