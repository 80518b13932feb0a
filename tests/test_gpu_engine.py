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


def test_train_with_graph_replay(dev, tmp_path):
    """config.use_graph: after two eager batches per (phase, shape) the zero_grad + forward + backward of a step is a
    replayed HIP graph (engine.GraphedStep).  Same loop, same decisions: both phases are walked, the counters agree with
    the eager run, the validation losses fall, a repeated run reproduces every number bit for bit (dropout 0, the
    uniform draws of the DLS stack come from torch's graph-safe generator re-seeded the same way), and the graphed run's
    losses agree with the eager run's to what a different random stream allows."""
    from pevae_amd import engine
    from pevae_amd.dataset import YELPDataLoader, build_vocab
    from pevae_amd.dls_vae import DLSVAE, default_config
    train_rows, valid_rows = _rows(192, 1, 60, 10), _rows(48, 2, 60, 10)
    vocab, rev, _ = build_vocab((train_rows, valid_rows), 100)

    def run(use_graph, tag):
        cfg = default_config(vocab_size=len(vocab), embed_size=32, dec_cell_size=64, max_seq_len=12, n_user=30, n_item=20,
                             latent_size=16, dropout=0.0, batch_size=32, init_lr=3e-3, grad_clip=5.0, seed=0,
                             teach_force_bound=1000, freeze_step=12, ckpt_step=6, print_step=3, lr_decay_rate=0.8,
                             improve_threshold=0.996, patient_increase=2.0, exp_name="t", fix_batch=False,
                             latent_stack="dls", use_graph=use_graph)
        torch.manual_seed(0)
        random.seed(0)
        np.random.seed(0)
        model = DLSVAE(cfg).to(dev)
        tf = YELPDataLoader("train", train_rows, cfg).attach(rev, dev)
        vf = YELPDataLoader("valid", valid_rows, cfg).attach(rev, dev)
        hist = engine.train(model, tf, vf, cfg, max_evals=4, checkpoint=os.path.join(str(tmp_path), tag + ".pt"))
        torch.cuda.synchronize()
        return hist, model

    h_graph, m_graph = run(True, "g1")
    h_again, _ = run(True, "g2")
    h_eager, _ = run(False, "e")
    assert len(h_graph) == 4 and m_graph.restarted
    assert [h[0] for h in h_graph] == [h[0] for h in h_eager]            # evaluations at the same batch counts
    assert h_graph[1][1] < h_graph[0][1]                                 # it learns
    assert h_graph == h_again                                            # replay is deterministic
    for (_, a, _), (_, b, _) in zip(h_graph, h_eager):
        assert abs(a - b) <= 0.05 * abs(b), (h_graph, h_eager)


def test_engine_skip_sequence_matches_reference_on_the_device(dev, tmp_path):
    """The real FlatAdam (guards and Adam on the device, skip flags read back asynchronously) under engine.train against
    the reference's control flow (oracle/ref_engine.py <- main_pretrain.py:148-371): the gradient norm of every step is
    scripted through the loss scale (spikes, a run of non-finite steps), the validation losses are scripted, and the
    sequence of skipped / applied steps, evaluations, learning rates and the freeze switch must agree."""
    from argparse import Namespace
    from pevae_amd import engine
    from pevae_amd.utils import Pack
    from oracle.ref_engine import ref_train_trace
    from test_engine_loop import Feed, script

    N = 1024
    norms, valid = script(7, 120)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for name in engine.FROZEN_IN_PHASE_2:
                setattr(self, name, torch.nn.Linear(2, 2))
            self.w = torch.nn.Parameter(torch.zeros(N))
            self.n_step, self.n_eval, self.restarted = 0, 0, False

        def forward(self, batch, mode=None):
            if not self.training:
                return Pack(vae_nll=torch.tensor(valid[self.n_eval], device=dev))
            scale = float(norms[self.n_step]) / float(np.sqrt(N))         # |grad| = scale * sqrt(N)
            self.n_step += 1
            return Pack(vae_nll=(self.w * scale).sum())

        def backward(self, batch_cnt, loss, step=None):
            loss.vae_nll.backward()

    cfg = Namespace(init_lr=1e-3, grad_clip=5.0, batch_size=4, teach_force_bound=1000, freeze_step=20, ckpt_step=8,
                    print_step=5, lr_decay_rate=0.8, improve_threshold=0.996, patient_increase=2.0, exp_name="t",
                    warmup_steps=6)
    # what the device computes as the norm: sqrt(sum of N squares of a float32 scale), not the scripted value itself
    dev_norms = [float(np.sqrt(np.float32(N)) * np.float32(float(n) / float(np.sqrt(N)))) for n in norms]
    want, want_lr = ref_train_trace(dev_norms, valid, cfg, batches_per_epoch=13, max_evals=9)
    model = M().to(dev)
    trace = []
    engine.train(model, Feed("train", 13), Feed("valid", 2, model), cfg, max_evals=9,
                 checkpoint=os.path.join(str(tmp_path), "m.pt"), trace=trace)
    got = [e for e in trace if e[0] != "lr"]
    got_lr = [e[2] for e in trace if e[0] == "lr"]
    last = max(i for i, e in enumerate(want) if e[0] == "eval")
    assert got[:last + 1] == [tuple(e) for e in want[:last + 1]]
    np.testing.assert_allclose(got_lr[:len(want_lr)], want_lr[:len(got_lr)], rtol=1e-6)
    assert sum(1 for e in got if e[0] == "step" and e[2]) >= 2 and any(e[0] == "freeze" for e in got)


def test_freeze_phase_with_the_dls_stack_and_gen_ids(dev, tmp_path):
    """latent_stack='dls' through the freeze switch (its Block -> Block connector is `x_init_connector`, as in
    dialog_models.py:3410, and is frozen with the sentence side), and free-running decoding on the real data path: the
    decoder starts from rev_vocab['<s>'] and lengths stop at rev_vocab['</s>'] (dialog_models.py:455-456)."""
    from pevae_amd import engine
    from pevae_amd.dataset import YELPDataLoader, build_vocab, BOS, EOS
    from pevae_amd.dls_vae import DLSVAE, config_from_vocab, default_config
    from pevae_amd.enc2dec.decoders import GEN
    train_rows, valid_rows = _rows(96, 3, 40, 8), _rows(32, 4, 40, 8)
    vocab, rev, _ = build_vocab((train_rows, valid_rows), 100)
    cfg = default_config(embed_size=32, dec_cell_size=64, max_seq_len=12, n_user=30, n_item=20,
                         latent_size=16, dropout=0.0, batch_size=32, init_lr=3e-3, grad_clip=5.0, seed=0,
                         teach_force_bound=1000, freeze_step=4, ckpt_step=3, print_step=3, lr_decay_rate=0.8,
                         improve_threshold=0.996, patient_increase=2.0, exp_name="t2", fix_batch=False, latent_stack="dls")
    config_from_vocab(cfg, rev)
    assert (cfg.bos_id, cfg.eos_id) == (rev[BOS], rev[EOS]) and {cfg.bos_id, cfg.eos_id} == {4, 5}
    torch.manual_seed(0)
    model = DLSVAE(cfg).to(dev)
    assert "x_init_connector.0.c1.weight_orig" in model.state_dict()
    tf = YELPDataLoader("train", train_rows, cfg).attach(rev, dev)
    vf = YELPDataLoader("valid", valid_rows, cfg).attach(rev, dev)
    hist = engine.train(model, tf, vf, cfg, max_evals=3, checkpoint=os.path.join(str(tmp_path), "b.pt"))
    assert len(hist) == 3 and model.restarted
    frozen = [p for n in engine.FROZEN_IN_PHASE_2 for p in getattr(model, n).parameters()]
    assert frozen and all(not p.requires_grad for p in frozen)
    assert any(p.requires_grad for p in model.sup_decoder.parameters())
    # GEN on a real batch: first decoder input and the EOS the length logic looks for
    model.eval()
    seen = []
    orig = model.x_decoder._hidden_states
    model.x_decoder._hidden_states = lambda inp, *a, **k: (seen.append(inp.clone()), orig(inp, *a, **k))[1]
    batch = tf.batcher(rev, cfg.max_seq_len, dev)(train_rows[:8])
    with torch.no_grad():
        model.backLoss = False
        ret, _ = model(batch, mode=GEN)
        model.backLoss = True
    assert seen and bool((seen[0] == rev[BOS]).all())
    seq = torch.cat(ret["sequence"], 1).cpu()
    for b, ln in enumerate(ret["length"]):
        eos_at = (seq[b] == rev[EOS]).nonzero()
        assert ln == (int(eos_at[0]) + 1 if len(eos_at) else seq.shape[1])


def test_recurrence_timeout_inside_a_replayed_graph_is_reported(dev):
    """a captured GRU launch zeroes its own sync buffer (error word included) at the start of every replay and no
    Python runs during a replay, so a wait that gave up in replay n was wiped by replay n + 1 and training went on with
    garbage states.  The kernels now also set a sticky process-level word that no launch clears; ops.gru_check_all (every
    evaluation of engine.train / validate) raises on it.  Here the time-out is injected by the library's one-thread test
    kernel INSIDE a captured graph that is followed by real recurrence launches and replayed several times."""
    from pevae_amd import _lib, ops
    ops.gru_check_all()                                    # clean start (also clears the sticky word)
    B, L, E, H = 16, 5, 12, 32
    torch.manual_seed(0)
    x = torch.randn(B, L, E, device=dev)
    h0 = torch.zeros(B, H, device=dev)
    w_ih, w_hh = torch.randn(3 * H, E, device=dev) * 0.1, torch.randn(3 * H, H, device=dev) * 0.1
    b = torch.zeros(3 * H, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.no_grad():
            for _ in range(2):
                ops.GRULayer.apply(x, h0, w_ih, w_hh, b, b, False)
        side.synchronize()
        ops.gru_check_all()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            _lib.call("pv_rnn_sync_error_inject", ops._stream())
            with torch.no_grad():
                out = ops.GRULayer.apply(x, h0, w_ih, w_hh, b, b, False)[0]
        ops.LAST_GRU_SYNC.clear()                          # what engine.GraphedStep does after a capture
        for _ in range(3):
            graph.replay()                                 # every replay re-zeroes the launch's own error word
        side.synchronize()
    assert torch.isfinite(out).all()
    with pytest.raises(_lib.PevaeHipError, match="timed out"):
        ops.gru_check_all()
    ops.gru_check_all()                                    # the check cleared the word: the next one passes


def test_skip_flags_report_their_own_step_when_the_host_runs_ahead(dev):
    """FlatAdam.skip_flag_async hands out handles on a ring of 8 pinned slots.  A host that runs more than 8
    steps ahead of its reads (graph replay with no report boundary in reach) used to re-arm a slot's event and overwrite
    its buffer while the handle of 8 steps earlier was still unread: that handle then reported the LATER step.  A slot now
    resolves its outstanding handle before it is reused: 20 handles read only after the last step each tell their own
    step's decision and learning rate."""
    from pevae_amd.optim import FlatAdam
    p = torch.nn.Parameter(torch.ones(64, device=dev))
    opt = FlatAdam([p], lr=1e-2, max_norm=5.0, warmup_steps=10)
    bad_steps = {3, 11, 12}
    flags = []
    for step in range(20):
        opt.zero_grad()
        p.grad.fill_(float("inf") if step in bad_steps else 0.1)
        opt.step()
        flags.append(opt.skip_flag_async())               # never read inside the loop
    got = [f.skipped() for f in flags]
    assert got == [s in bad_steps for s in range(20)]
    # the warm-up schedule counts APPLIED steps: the k-th applied step runs at lr * min(1, k / warmup)
    applied = 0
    for s, f in enumerate(flags):
        if s in bad_steps:
            continue
        assert abs(f.lr() - 1e-2 * min(1.0, applied / 10.0)) <= 1e-9, (s, f.lr())
        applied += 1
