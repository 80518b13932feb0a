"""engine.train at the Yelp shape (bs=256, L=50, V=20004, DLS stack) on resident synthetic batches, eager vs
config.use_graph: steps per second of the real training loop (guards, skip bookkeeping, reports included), over the whole call and from
step 50 on (host clock at next_batch; the loop cannot run more than 8 steps ahead of the device).
python tools/bench_engine.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pevae_amd import engine  # noqa: E402
from pevae_amd.dls_vae import DLSVAE, default_config, synthetic_batch  # noqa: E402


class Feed(object):
    def __init__(self, name, batches, per_epoch):
        self.name, self.batches, self.per_epoch, self.i = name, batches, per_epoch, 0
        self.stamps = {}

    def epoch_init(self, config, shuffle=True, verbose=False):
        self.i = 0

    def next_batch(self):
        if self.i >= self.per_epoch:
            return None
        self.i += 1
        if self.i in (50, self.per_epoch):            # the host is never more than 8 steps ahead (skip-flag ring)
            self.stamps[self.i] = time.time()
        return self.batches[self.i % len(self.batches)]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dev = torch.device("cuda:0")
    for use_graph in (False, True):
        cfg = default_config(dropout=0.3, batch_size=256, latent_stack="dls", teach_force_bound=10 ** 6, freeze_step=None,
                             ckpt_step=steps, print_step=100, lr_decay_rate=0.5, improve_threshold=0.996, patient_increase=2.0,
                             exp_name="bench_engine", use_graph=use_graph)
        torch.manual_seed(0)
        model = DLSVAE(cfg).to(dev)
        train = Feed("train", [synthetic_batch(cfg, 256, 100 + i, dev) for i in range(4)], steps)
        valid = Feed("valid", [synthetic_batch(cfg, 256, 900, dev)], 1)
        torch.cuda.synchronize()
        t0 = time.time()
        engine.train(model, train, valid, cfg, max_evals=1, checkpoint="/tmp/bench_engine.pt")
        torch.cuda.synchronize()
        dt = time.time() - t0
        steady = (train.stamps[steps] - train.stamps[50]) / (steps - 50)
        print("use_graph=%s: %d steps + 1 evaluation in %.2f s -> %.2f ms/step over the whole call (warm-up steps, capture, "
              "evaluation and checkpoint included), %.2f ms/step = %.0f samples/s from step 50 on" %
              (use_graph, steps, dt, dt / steps * 1e3, steady * 1e3, 256 / steady))


if __name__ == "__main__":
    main()
