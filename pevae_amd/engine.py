"""Training / validation loop of the DLS_VAE step: the control flow of main_pretrain.train / validate
(models/main_pretrain.py:148-399) around the HIP step.

Kept from the reference: the epoch / batch iteration over a feed with epoch_init / next_batch, TEACH_FORCE before
`teach_force_bound` evaluations and GEN after (:200-203), `model.backward` on the selected loss (:205), global-norm clip
and the two skip rules (:207-225; evaluated on the device inside FlatAdam, same arithmetic), the freeze phase (:227-266:
reload the best checkpoint, freeze embedding / encoder / decoder / init connector, fresh optimizer), evaluation every
`ckpt_step` batches with plateau decay `lr *= lr_decay_rate` when the validation loss did not improve by 1 % (:343-348),
patience bookkeeping and best-checkpoint save (:354-366), LossManager's report format (:21-73), the skip bookkeeping of
:213-225 (a skipped step adds nothing to train_loss, does not advance batch_cnt and triggers no report, evaluation or
freeze switch -- see train()), AdamW for the freeze phase (dialog_models.py:834).
Changed: LossManager keeps the per-batch losses on the device and reads them back once per report instead of one
`.item()` (a host sync) per loss per batch; TensorBoard, generation previews and the text metrics (:264-265, :319-338,
:401-1286) are out of scope."""
import logging
import os
from collections import defaultdict

import numpy as np
import torch

from .enc2dec.decoders import GEN, TEACH_FORCE
from .optim import FlatAdam

logger = logging.getLogger("pevae_amd.engine")


class LossManager(object):
    """main_pretrain.py:21-73 with device-side accumulation: add_loss appends 0-d device tensors; the averages are
    formed (one stacked read-back per key) only when a report asks for them."""

    def __init__(self):
        self.losses = defaultdict(list)
        self.backward_losses = []

    def add_loss(self, loss):
        for key, val in loss.items():
            if val is not None and type(val) is not bool and type(val) is not int:
                self.losses[key].append(val.detach() if torch.is_tensor(val) else torch.tensor(float(val)))

    def add_backward_loss(self, loss):
        self.backward_losses.append(loss.detach())

    def clear(self):
        self.losses = defaultdict(list)
        self.backward_losses = []

    def _avg(self, vals, window):
        vals = vals if window is None else vals[-window:]
        return float(torch.stack([v.reshape(()).float() for v in vals]).mean().item())

    def return_dict(self, window=None):
        ret = {}
        for key, vals in self.losses.items():
            if not vals:
                continue
            avg = self._avg(vals, window)
            ret[key] = avg
            if "nll" in key and "PPL" not in self.losses:
                ret[key.split("nll")[0] + "PPL"] = float(np.exp(avg))
        return ret

    def pprint(self, name, window=None, prefix=None):
        parts = []
        for key, vals in self.losses.items():
            if not vals:
                continue
            avg = self._avg(vals, window)
            parts.append("{} {:.3f}".format(key, avg))
            if "nll" in key and "PPL" not in self.losses:
                parts.append("PPL {:.3f}".format(np.exp(avg)))
        body = "{} {}".format(name, " ".join(parts))
        return "{}: {}".format(prefix, body) if prefix else body

    def avg_loss(self):
        return self._avg(self.losses["vae_nll"], None)


def adjust_learning_rate(optimizer, last_lr, decay_rate=0.5):           # main_pretrain.py:75-80
    lr = last_lr * decay_rate
    optimizer.lr = optimizer.lr * decay_rate
    return lr


def validate(model, valid_feed, config, batch_cnt=None):                # main_pretrain.py:373-399
    model.eval()
    with torch.no_grad():
        valid_feed.epoch_init(config, shuffle=True, verbose=True)
        losses = LossManager()
        while True:
            batch = valid_feed.next_batch()
            if batch is None:
                break
            losses.add_loss(model(batch, mode=TEACH_FORCE))
        valid_loss = losses.avg_loss()
        logger.info(losses.pprint(valid_feed.name))
        logger.info("Total valid loss {}".format(valid_loss))
        res = losses.return_dict()
    _check_recurrence()
    return valid_loss, res


FROZEN_IN_PHASE_2 = ("x_embedding", "x_encoder", "x_decoder", "x_init_connector")    # main_pretrain.py:238-247


def make_optimizer(model, config, phase2=False):
    """model.get_optimizer (dialog_models.py:122-131: Adam + constant-with-warmup) or, for the freeze phase,
    model.get_freeze_optimizer (:834: torch.optim.AdamW with its default decoupled weight decay 0.01)."""
    return FlatAdam([p for p in model.parameters() if p.requires_grad], lr=config.init_lr, max_norm=config.grad_clip,
                    warmup_steps=getattr(config, "warmup_steps", 0), weight_decay=0.01 if phase2 else 0.0)


class GraphedStep(object):
    """zero_grad + forward + backward of one training batch replayed from a HIP graph (config.use_graph).

    An eager step pays ~230 ctypes calls and as many launches' worth of host work; the device needs 3.5 ms for it.  The
    first `warmup` batches of every (mode, phase, batch shape) run eagerly -- they are real steps: they allocate every
    workspace and take the lazy paths once --, the next one is CAPTURED (torch.cuda.graph) on static copies of its
    tensors and then replayed, and from then on a step is: copy the batch into the static tensors, one graph launch.
    The guards / clip / Adam of the step are part of the graph too (`with_step`): the base learning rate is a device scalar
    that the host writes at evaluations (FlatAdam.lr), the decisions are taken on the device, and the skip flag is read back
    as before.  Only a batch that could be the freeze phase's switching batch -- whose gradients the reference discards after
    looking at the guards (:207-266) -- runs the graph WITHOUT the optimizer step and lets the loop decide.  New
    dropout masks every replay: the generator's step counter lives in device memory and is advanced inside the graph;
    torch's own generator (the uniform draws of DLS_Auto) is graph-safe.  The loss Pack handed back holds CLONES of the
    static scalars (the next replay overwrites those).  Batches of another shape (a ragged last batch) or in GEN mode run
    eagerly.  Everything -- warm-up, capture, replay and the eager fall-backs -- must run on ONE non-default stream
    (autograd ties a node to the stream it was created on; the legacy default stream cannot take part in a capture):
    train() enters `self.stream` for the whole loop."""

    def __init__(self, model, warmup=2):
        self.model, self.warmup = model, warmup
        self.stream = torch.cuda.Stream()
        self.seen, self.graphs = defaultdict(int), {}
        self.pool = None

    def invalidate(self):
        """The set of trained parameters changed (freeze phase): every captured graph is stale."""
        self.seen.clear()
        self.graphs.clear()

    @staticmethod
    def _key(batch, mode, model):
        sig = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(batch.items()) if torch.is_tensor(v))
        return (mode, bool(getattr(model, "restarted", False)), sig)

    def _eager(self, optimizer, batch, mode, batch_cnt, with_step=False):
        if hasattr(optimizer, "flat_g"):
            optimizer.zero_grad(beside_forward=True)       # FlatAdam: the fill runs beside the head of the forward pass
        else:
            optimizer.zero_grad()
        loss = self.model(batch, mode=mode)
        self.model.backward(batch_cnt, loss, step=batch_cnt)
        if with_step:
            optimizer.step()                      # clip, skip rules and Adam on the device (:207-225, :268-269)
        return loss

    def __call__(self, optimizer, batch, mode, batch_cnt, with_step=False):
        """-> loss Pack.  with_step: optimizer.step() is part of what runs (and of what is captured); the caller then only
        reads the skip flag back."""
        tensors = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        if mode != TEACH_FORCE or not tensors or not all(v.is_cuda for v in tensors.values()):
            return self._eager(optimizer, batch, mode, batch_cnt, with_step)
        key = self._key(batch, mode, self.model) + (bool(with_step), id(optimizer))
        entry = self.graphs.get(key)
        if entry is None:
            self.seen[key] += 1
            if self.seen[key] <= self.warmup:
                return self._eager(optimizer, batch, mode, batch_cnt, with_step)
            import gc
            gc.collect()                                   # no autograd graph of an eager step may be alive during capture
            static = dict(batch)
            static.update({k: v.clone() for k, v in tensors.items()})
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self.pool, stream=self.stream, capture_error_mode="thread_local"):
                out = self._eager(optimizer, static, mode, batch_cnt, with_step)
            if self.pool is None:
                self.pool = graph.pool()
            entry = self.graphs[key] = (graph, static, out)
            from . import ops
            # The sync buffers of captured launches are rewritten by every replay, so their error words say nothing about
            # earlier replays: timed-out waits of replayed recurrence launches are reported through the library's sticky
            # word instead (pv_rnn_sync_error, read by ops.gru_check_all at every evaluation).
            ops.LAST_GRU_SYNC.clear()
        graph, static, out = entry
        for k, v in tensors.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return type(out)({k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in out.items()})


def _check_recurrence():
    """A timed-out wait inside a persistent GRU / LSTM launch is reported through an error word only; read the words of
    the recent eager launches AND the library's sticky word (which also covers launches replayed from HIP graphs) at every
    evaluation so that a run cannot continue on garbage states (one small read-back)."""
    if torch.cuda.is_available():
        from . import ops
        ops.gru_check_all()


def train(model, train_feed, valid_feed, config, max_evals=None, checkpoint=None, optimizer_factory=None, trace=None):
    """main_pretrain.train (:148-371); see _train_loop.  config.use_graph (GPU only): the zero_grad + forward + backward
    of every regular batch is replayed from a HIP graph (GraphedStep) and the whole loop runs on that object's stream."""
    if getattr(config, "use_graph", False) and torch.cuda.is_available():
        graphed = GraphedStep(model)
        torch.cuda.current_stream().synchronize()
        with torch.cuda.stream(graphed.stream):
            history = _train_loop(model, train_feed, valid_feed, config, max_evals, checkpoint, optimizer_factory, trace, graphed)
            graphed.stream.synchronize()
        return history
    return _train_loop(model, train_feed, valid_feed, config, max_evals, checkpoint, optimizer_factory, trace, None)


def _train_loop(model, train_feed, valid_feed, config, max_evals, checkpoint, optimizer_factory, trace, graphed):
    """Runs until the reference's stop rule (done_epoch > 100, :184; five evaluations after the freeze phase add 100,
    :333-334) or `max_evals` evaluations.  Returns the history of (batch_cnt, valid_loss, lr) per evaluation.
    config fields used (the reference's names): init_lr, grad_clip, batch_size, teach_force_bound, freeze_step,
    ckpt_step, print_step, lr_decay_rate, improve_threshold, patient_increase, exp_name.

    Skipped steps (:213-222 `continue`): the decision is taken on the device inside optimizer.step(); the host needs it
    for the same bookkeeping as the reference -- a skipped step adds nothing to train_loss and does not advance
    batch_cnt, so it triggers no report, evaluation or freeze switch.  Each step's flag comes back through an 8-byte
    asynchronous copy (FlatAdam.skip_flag_async); steps whose flag has not been read are `pending`.  The host waits for
    the flags only when one of the pending steps could reach a print / evaluation / freeze boundary, which keeps the
    reference's exact sequence without a host sync per step (the reference has three).
    optimizer_factory(model, config, phase2) and `trace` (a list that receives ("step", n, skipped) / ("eval", batch_cnt,
    valid_loss, lr) / ("freeze", n) events) exist for the CPU test against oracle/ref_engine.py."""
    factory = optimizer_factory or make_optimizer
    patience = 2
    valid_loss_threshold = np.inf
    best_valid_loss = np.inf
    learning_rate = config.init_lr
    batch_cnt = 0
    optimizer = factory(model, config, False)
    best_model_name = checkpoint or os.path.join("./runs/", "model_" + getattr(config, "exp_name", "pevae") + ".pt")
    os.makedirs(os.path.dirname(os.path.abspath(best_model_name)), exist_ok=True)
    done_epoch, final_epoch, evals, nstep = 0, 0, 0, 0
    history = []
    train_loss = LossManager()
    model.train()
    model.restarted = getattr(model, "restarted", False)
    freeze_step = getattr(config, "freeze_step", None)
    pending = []                                      # [(loss pack, SkipFlag, step number)] in step order

    def note(*ev):
        if trace is not None:
            trace.append(ev)

    def boundary_within(lo, hi):
        """Could batch_cnt, now `lo`, reach a print / evaluation / freeze boundary while growing to at most `hi`?"""
        if hi <= lo:
            return False
        for step_ in (config.print_step, config.ckpt_step):
            if hi // step_ != lo // step_:
                return True
        return freeze_step is not None and not model.restarted and hi >= freeze_step

    def settle(block):
        """Fold the steps whose skip flag has arrived (all of them when block) into batch_cnt / train_loss."""
        nonlocal batch_cnt
        last_applied = None
        while pending and (block or pending[0][1].ready()):
            loss, flag, n = pending.pop(0)
            skipped = flag.skipped()
            note("step", n, skipped)
            last_applied = not skipped
            if skipped:
                logger.info("step %d skipped by the gradient-norm guard" % n)
                continue
            if trace is not None:
                trace.append(("lr", n, flag.lr()))
            train_loss.add_loss(loss)
            batch_cnt += 1
        return last_applied

    logger.info("**** Training Begins ****")
    while done_epoch <= 100 and (max_evals is None or evals < max_evals):
        train_feed.epoch_init(config, shuffle=True)
        while max_evals is None or evals < max_evals:
            batch = train_feed.next_batch()
            if batch is None:
                break
            model.steps = batch_cnt + len(pending)
            mode = TEACH_FORCE if done_epoch < config.teach_force_bound else GEN
            # could this batch be the one at which the freeze phase starts?  (its gradients are then only looked at, :207-266)
            may_switch = freeze_step is not None and not model.restarted and batch_cnt + len(pending) >= freeze_step
            stepped = False
            if graphed is not None:
                stepped = not may_switch and hasattr(optimizer, "skip_flag_async")
                loss = graphed(optimizer, batch, mode, batch_cnt, with_step=stepped)
            else:
                optimizer.zero_grad()
                loss = model(batch, mode=mode)
                model.backward(batch_cnt, loss, step=batch_cnt)
            nstep += 1
            if may_switch:
                settle(True)
                if batch_cnt >= freeze_step:
                    # :207-266: the guards come first -- a skipped step `continue`s before the switch is looked at
                    optimizer.guard()
                    flag = optimizer.skip_flag_async()
                    if flag.skipped():
                        note("step", nstep, True)
                        continue
                    logger.info("Switch to another train!")
                    note("freeze", nstep)
                    best_valid_loss = np.inf
                    if os.path.exists(best_model_name):
                        model.load_state_dict(torch.load(best_model_name, weights_only=True), strict=False)
                        logger.info("Model {}: Load Done!".format(best_model_name))
                    model.restarted = True
                    for name in FROZEN_IN_PHASE_2:
                        mod = getattr(model, name, None)
                        if mod is None:                  # a renamed module would silently keep training
                            raise AttributeError("engine.train: the model has no %r to freeze (main_pretrain.py:238-247)" % name)
                        for p in mod.parameters():
                            p.requires_grad = False
                            p.grad = None
                    optimizer = factory(model, config, True)             # :257 get_freeze_optimizer (AdamW)
                    if graphed is not None:
                        graphed.invalidate()
                    learning_rate = config.init_lr
                    batch_cnt = 0
                    continue
            if not stepped:
                optimizer.step()                   # clip, skip rules and Adam on the device (:207-225, :268-269)
            pending.append((loss, optimizer.skip_flag_async(), nstep))
            applied = settle(boundary_within(batch_cnt, batch_cnt + len(pending)))
            if pending or not applied:             # no boundary in reach, or this step was skipped (`continue`, :216 / :222)
                continue
            if batch_cnt % config.print_step == 0:
                logger.info(train_loss.pprint("Train", window=config.print_step,
                                              prefix="{}/{}".format(batch_cnt % (config.ckpt_step + 1), config.ckpt_step)))
            if batch_cnt % config.ckpt_step == 0:
                optimizer.zero_grad()
                logger.info("\n=== Evaluating Model ===")
                logger.info(train_loss.pprint("Train"))
                done_epoch += 1
                evals += 1
                _check_recurrence()
                valid_loss, _ = validate(model, valid_feed, config, batch_cnt)
                if model.restarted:
                    final_epoch += 1
                    if final_epoch >= 5:
                        done_epoch += 100
                if learning_rate > 1e-6 and valid_loss > best_valid_loss * 0.99:             # :343-348
                    learning_rate = adjust_learning_rate(optimizer, learning_rate, config.lr_decay_rate)
                    logger.info("Adjust learning rete to {}".format(learning_rate))
                if valid_loss <= valid_loss_threshold * config.improve_threshold:             # :354-366
                    patience = max(patience, done_epoch * config.patient_increase)
                    valid_loss_threshold = valid_loss
                    if not model.restarted:
                        torch.save(model.state_dict(), best_model_name)
                    best_valid_loss = valid_loss
                history.append((batch_cnt, valid_loss, learning_rate))
                note("eval", batch_cnt, valid_loss, learning_rate)
                model.train()
                train_loss.clear()
                logger.info("\n**** Epoch {} ****".format(done_epoch))
    settle(True)
    return history
