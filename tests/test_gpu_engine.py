"""Engine loop (pevae_amd/engine.py, main_pretrain.train / validate semantics) over the Yelp data path on the device:
a few evaluations of a small model through both training phases."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rows(n, seed, n_words, max_tok):
    from pevae_amd.dataset import BOS, EOS
    rnd = random.Random(seed)
    words = ["w%d" % i for i in range(n_words)]
    out = []
    for _ in range(n):
        u, it = rnd.randrange(30), rnd.randrange(20)
        k = rnd.randint(2, max_tok)
        # the sentence depends on (user, item): something for the conditional model to learn
        toks = [BOS] + [words[(u * 7 + it * 3 + j * (1 + u % 3)) % n_words] for j in range(k)] + [EOS]
        out.append((u, it, 0, toks, rnd.randint(1, 5), rnd.gauss(0, 1)))
    return out


def test_train_validate_freeze_phase(dev, tmp_path):
    from pevae_amd import engine
    from pevae_amd.dataset import YELPDataLoader, build_vocab, BOS, EOS
    from pevae_amd.dls_vae import DLSVAE, default_config
    train_rows, valid_rows = _rows(192, 1, 60, 10), _rows(48, 2, 60, 10)
    vocab, rev, _ = build_vocab((train_rows, valid_rows), 100)
    assert rev[BOS] == 4 or rev[EOS] == 4                   # the two most frequent tokens follow the four specials
    cfg = default_config(vocab_size=len(vocab), embed_size=32, dec_cell_size=64, max_seq_len=12, n_user=30, n_item=20,
                         latent_size=16, dropout=0.0, batch_size=32, init_lr=3e-3, grad_clip=5.0, seed=0,
                         teach_force_bound=1000, freeze_step=12, ckpt_step=6, print_step=3, lr_decay_rate=0.8,
                         improve_threshold=0.996, patient_increase=2.0, exp_name="t", fix_batch=False)
    torch.manual_seed(0)
    model = DLSVAE(cfg).to(dev)
    tf = YELPDataLoader("train", train_rows, cfg).attach(rev, dev)
    vf = YELPDataLoader("valid", valid_rows, cfg).attach(rev, dev)
    b = tf.batcher(rev, cfg.max_seq_len, dev)(train_rows[:5])
    assert b["input_ids"].is_cuda and b["input_ids"].dtype == torch.long and b["ratings"].dtype == torch.float32
    ckpt = os.path.join(str(tmp_path), "best.pt")
    first_valid, _ = engine.validate(model, vf, cfg)
    model.train()
    hist = engine.train(model, tf, vf, cfg, max_evals=4, checkpoint=ckpt)
    assert len(hist) == 4 and all(np.isfinite(h[1]) for h in hist)
    assert hist[1][1] < first_valid                         # it learns: validation NLL falls in phase 1
    assert os.path.exists(ckpt)                             # best checkpoint written before the freeze phase
    assert model.restarted                                  # batch_cnt reached freeze_step: phase 2 entered
    assert hist[2][0] == 6                                  # ... with the batch counter restarted
    frozen = [p for n in engine.FROZEN_IN_PHASE_2 for p in getattr(model, n).parameters()]
    assert frozen and all(not p.requires_grad for p in frozen)
    assert any(p.requires_grad for p in model.parameters())
    # a frozen parameter keeps the value of the reloaded checkpoint through phase 2
    sd = torch.load(ckpt, weights_only=True)
    assert torch.equal(model.x_embedding.weight.detach().cpu(), sd["x_embedding.weight"].cpu())
    assert all(h[2] <= cfg.init_lr for h in hist)
