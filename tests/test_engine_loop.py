"""Host logic of the training loop (pevae_amd/engine.py) against the restatement of the reference's control flow
(oracle/ref_engine.py <- models/main_pretrain.py:148-371): which steps are skipped, which advance batch_cnt, when the
evaluations and the freeze switch happen, and what the learning rate is at each of them -- on scripted gradient norms
(spikes, inf / nan runs) and scripted validation losses.  CPU: the optimizer is a scripted stand-in that takes the guard
decision the way the device kernel does (csrc/optim.hip guard_kernel) and hands the flag back late, so the loop's
deferred bookkeeping is exercised; tests/test_gpu_engine.py runs the same scripts through the real FlatAdam."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from pevae_amd import engine
from pevae_amd.utils import Pack
from oracle.ref_engine import ref_train_trace


def script(seed, n):
    rnd = np.random.RandomState(seed)
    norms = list(np.abs(rnd.normal(2.0, 0.3, n)).astype(np.float32))
    for at in rnd.choice(np.arange(3, n), size=n // 9, replace=False):       # isolated spikes
        norms[at] = np.float32(norms[at] * rnd.choice([3.5, 8.0, 40.0]))
    start = int(rnd.randint(5, n - 10))
    for k in range(7):                                                       # 7 non-finite in a row: 5 skipped, the 6th is not
        norms[start + k] = np.float32(np.inf if k % 2 else np.nan)
    valid = [float(np.float32(v)) for v in 5.0 - 0.4 * np.arange(40) + rnd.normal(0, 0.5, 40)]     # exact in float32
    return norms, valid


class Flag(object):
    def __init__(self, skip, lr, late):
        self.skip, self._lr, self.late = skip, lr, late

    def ready(self):
        return not self.late

    def skipped(self):
        return self.skip

    def lr(self):
        return self._lr


class ScriptedOptimizer(object):
    """Guard state machine of csrc/optim.hip guard_kernel in Python, fed with the scripted norms."""

    def __init__(self, feed_norm, lr, warmup, late):
        self.feed_norm, self.lr, self.warmup, self.late = feed_norm, lr, warmup, late
        self.avg, self.have, self.jump, self.cnt, self.k = np.float32(0), False, 0, 0, 0
        self.skip, self.lr_eff = False, 0.0

    def zero_grad(self):
        pass

    def guard(self):
        f32 = np.float32
        norm = f32(self.feed_norm())
        self.skip = False
        if not self.have:
            self.avg, self.have = norm, True
        if (np.isinf(norm) or np.isnan(norm)) and self.jump < 5:
            self.jump += 1
            self.skip = True
        else:
            self.jump = 0
            if self.avg * f32(3) < norm:
                self.avg = self.avg / f32(20) * f32(19) + norm / f32(20)
                self.skip = True
            else:
                self.avg = self.avg / f32(self.cnt + 1) * f32(self.cnt) + norm / f32(self.cnt + 1)
        if not self.skip:
            self.lr_eff = self.lr * self.k / max(1.0, self.warmup) if (self.warmup > 0 and self.k < self.warmup) else self.lr
            self.k += 1
            self.cnt += 1

    def apply(self):
        pass

    def step(self):
        self.guard()

    def skip_flag_async(self):
        return Flag(self.skip, self.lr_eff, self.late)


class Model(nn.Module):
    def __init__(self, valid_losses, norms):
        super().__init__()
        for name in engine.FROZEN_IN_PHASE_2:
            setattr(self, name, nn.Linear(2, 2))
        self.free = nn.Linear(2, 2)
        self.valid_losses, self.norms, self.n_eval, self.n_step, self.restarted = valid_losses, norms, 0, 0, False

    def forward(self, batch, mode=None):
        if self.training:
            self.current_norm = self.norms[self.n_step]
            self.n_step += 1
            return Pack(vae_nll=torch.tensor(1.0))
        return Pack(vae_nll=torch.tensor(float(self.valid_losses[self.n_eval])))

    def backward(self, batch_cnt, loss, step=None):
        pass


class Feed(object):
    def __init__(self, name, n, model=None):
        self.name, self.n, self.model, self.i = name, n, model, 0

    def epoch_init(self, config, shuffle=True, verbose=True):
        self.i = 0
        if self.name == "valid":
            self.pending_eval = True

    def next_batch(self):
        if self.i >= self.n:
            if self.name == "valid" and self.model is not None:
                self.model.n_eval += 1
            return None
        self.i += 1
        return {}


@pytest.mark.parametrize("late", [True, False])
@pytest.mark.parametrize("seed,freeze,warmup", [(0, 25, 0), (1, 40, 10), (2, None, 4), (3, 7, 0)])
def test_engine_trace_matches_reference_control_flow(tmp_path, late, seed, freeze, warmup):
    norms, valid = script(seed, 160)
    cfg = Namespace(init_lr=1e-3, grad_clip=5.0, batch_size=4, teach_force_bound=1000, freeze_step=freeze, ckpt_step=8,
                    print_step=5, lr_decay_rate=0.8, improve_threshold=0.996, patient_increase=2.0, exp_name="t",
                    warmup_steps=warmup)
    want, want_lr = ref_train_trace(norms, valid, cfg, batches_per_epoch=13, max_evals=12)
    model = Model(valid, norms)
    made = []

    def factory(m, c, phase2):
        made.append(phase2)
        return ScriptedOptimizer(lambda: m.current_norm, c.init_lr, c.warmup_steps, late)

    trace = []
    hist = engine.train(model, Feed("train", 13), Feed("valid", 2, model), cfg, max_evals=12,
                        checkpoint=os.path.join(str(tmp_path), "m.pt"), optimizer_factory=factory, trace=trace)
    got = [e for e in trace if e[0] != "lr"]
    got_lr = [e[2] for e in trace if e[0] == "lr"]
    n_want = len([e for e in want if e[0] == "eval"])
    assert n_want >= 6 and len(hist) == n_want            # (the freeze phase ends the run after 5 evaluations, :333-334)
    # every event up to the last evaluation, in order: skipped / applied steps, the freeze switch, the evaluations
    last = max(i for i, e in enumerate(want) if e[0] == "eval")
    assert got[:last + 1] == [tuple(e) for e in want[:last + 1]]
    np.testing.assert_allclose(got_lr[:len(want_lr)], want_lr[:len(got_lr)], rtol=1e-6)
    assert any(e[0] == "step" and e[2] for e in got)                      # the scripts do contain skips
    assert made == ([False, True] if freeze is not None else [False])     # phase 2 gets its own (AdamW) optimizer
    if freeze is not None:
        assert model.restarted and all(not p.requires_grad for n in engine.FROZEN_IN_PHASE_2
                                       for p in getattr(model, n).parameters())
        assert all(p.requires_grad for p in model.free.parameters())


def test_freeze_raises_on_a_missing_module(tmp_path):
    norms, valid = script(5, 60)
    cfg = Namespace(init_lr=1e-3, grad_clip=5.0, batch_size=4, teach_force_bound=1000, freeze_step=3, ckpt_step=8,
                    print_step=5, lr_decay_rate=0.8, improve_threshold=0.996, patient_increase=2.0, exp_name="t",
                    warmup_steps=0)
    model = Model(valid, norms)
    del model.x_init_connector
    with pytest.raises(AttributeError, match="x_init_connector"):
        engine.train(model, Feed("train", 13), Feed("valid", 2, model), cfg, max_evals=3,
                     checkpoint=os.path.join(str(tmp_path), "m.pt"),
                     optimizer_factory=lambda m, c, p: ScriptedOptimizer(lambda: m.current_norm, c.init_lr, 0, False))
