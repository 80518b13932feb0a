"""CPU restatement of the reference's training-loop control flow (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows models/main_pretrain.py:148-371 line by line for everything that decides WHAT happens WHEN -- the guards on the
gradient norm (:207-225), the freeze switch (:227-266), optimizer / scheduler steps (:268-269), batch_cnt (:288), the
evaluation every ckpt_step batches (:299-371) with plateau decay (:343-348, adjust_learning_rate :75-80) and the stop rule
(:184, :333-334) -- with the model, the data and the optimizer replaced by what the caller scripts:
  norms[n]        the pre-clip gradient norm of the n-th processed batch (n = 0, 1, ...; may be inf / nan),
  valid_losses[e] the validation loss of the e-th evaluation.
The arithmetic on the running norm is done in float32, as the reference does it on a float32 tensor.  The result is the
event trace pevae_amd.engine.train(..., trace=[]) must reproduce:
  ("step", n + 1, skipped), ("freeze", n + 1), ("eval", batch_cnt, valid_loss, learning_rate).
"""
import numpy as np


def ref_train_trace(norms, valid_losses, config, batches_per_epoch, max_evals=None):
    f32 = np.float32
    trace = []
    valid_loss_threshold = np.inf                              # :153
    best_valid_loss = np.inf                                   # :154
    learning_rate = config.init_lr                             # :156
    batch_cnt = 0                                              # :157
    done_epoch, final_epoch, evals = 0, 0, 0                   # :165, :181
    avger_norm = None                                          # :175
    jump_cnt = 0                                               # :177
    restarted = False
    lr_now = config.init_lr                                    # optimizer.param_groups[..]['lr']
    freeze_step = getattr(config, "freeze_step", None)
    n = 0
    lr_applied = []                                            # learning rate of every applied optimizer step
    sched_step = 0                                             # LambdaLR counter of the current optimizer
    warmup = getattr(config, "warmup_steps", 0)
    while True:
        if done_epoch > 100 or (max_evals is not None and evals >= max_evals):          # :184
            break
        for _ in range(batches_per_epoch):                     # :186-191
            if max_evals is not None and evals >= max_evals:
                break
            if n >= len(norms):
                return trace, lr_applied
            norm_sum = f32(norms[n])                           # :207-208
            n += 1
            if avger_norm is None:                             # :210-211
                avger_norm = norm_sum
            if (np.isinf(norm_sum) or np.isnan(norm_sum)) and jump_cnt < 5:             # :213-216
                jump_cnt += 1
                trace.append(("step", n, True))
                continue
            jump_cnt = 0                                       # :217
            if avger_norm * f32(3) < norm_sum:                 # :219-222
                avger_norm = avger_norm / f32(20) * f32(19) + norm_sum / f32(20)
                trace.append(("step", n, True))
                continue
            avger_norm = avger_norm / f32(batch_cnt + 1) * f32(batch_cnt) + norm_sum / f32(batch_cnt + 1)   # :225
            if freeze_step is not None and batch_cnt >= freeze_step and not restarted:  # :227-266
                trace.append(("freeze", n))
                best_valid_loss = np.inf
                restarted = True
                learning_rate = config.init_lr
                lr_now = config.init_lr                        # :257 a fresh optimizer
                sched_step = 0
                batch_cnt = 0
                avger_norm = None
                continue
            # :268-269 optimizer.step(); scheduler.step()  (constant-with-warmup: factor min(1, k / warmup))
            lr_applied.append(lr_now * (min(1.0, sched_step / max(1.0, warmup)) if warmup > 0 else 1.0))
            sched_step += 1
            trace.append(("step", n, False))
            batch_cnt += 1                                     # :288
            if batch_cnt % config.ckpt_step == 0:              # :299
                done_epoch += 1                                # :305
                evals += 1
                valid_loss = valid_losses[evals - 1]           # :308
                if restarted:                                  # :322-334
                    final_epoch += 1
                    if final_epoch >= 5:
                        done_epoch += 100
                if learning_rate > 1e-6 and valid_loss > best_valid_loss * 0.99:        # :343-348
                    learning_rate = learning_rate * config.lr_decay_rate                # :75-80
                    lr_now = lr_now * config.lr_decay_rate
                if valid_loss <= valid_loss_threshold * config.improve_threshold:       # :354-366
                    valid_loss_threshold = valid_loss
                    best_valid_loss = valid_loss               # (and the checkpoint save, :359-360)
                trace.append(("eval", batch_cnt, valid_loss, learning_rate))
    return trace, lr_applied
