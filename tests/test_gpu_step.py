"""Composed DLS_VAE training step (pevae_amd/dls_vae.py) on the HIP kernels vs the CPU oracle of the same
composition (oracle/ref_step.py): losses, every parameter gradient, and one clip + Adam update."""
import numpy as np
import pytest
import torch

from oracle.ref_step import RefDLSVAE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


# Full-size tolerances against the bf16-emulating oracle (chosen from PEVAE_TOL_REPORT=1 runs, about 3x the worst
# parameter): what is left is accumulation order, the recurrences' fast sigmoid / tanh, and elements that round to a
# different bf16 neighbour on the two sides.
FULL_LOSS_RTOL = 2e-3     # losses (worst seen: < 5e-4)
FULL_GRAD_TOL = 4e-3      # max |g - g_oracle| / max |g_oracle| per parameter (worst seen: 1.2e-3; was 8e-2 against the fp32 oracle)
FULL_COS_TOL = 5e-5       # 1 - cosine per parameter of >= 4096 elements (worst seen: 6e-6; was 1e-2)


class bf16_oracle(object):
    """Run the oracle with the matrix core's operand rounding (oracle.ref_modules.EMULATE_BF16_MFMA): the full-size
    comparisons then measure the kernels, not the bf16 format."""

    def __enter__(self):
        import oracle.ref_modules as R
        self.R, self.prev = R, R.EMULATE_BF16_MFMA
        R.EMULATE_BF16_MFMA = True

    def __exit__(self, *exc):
        self.R.EMULATE_BF16_MFMA = self.prev
        return False


def grad_report(name, g, r, worst):
    """max |g - r| / max |r| and 1 - cos, collected for PEVAE_TOL_REPORT=1 runs (how the tolerances below were chosen)."""
    scale = float(r.abs().max()) + 1e-9
    err = float((g - r).abs().max()) / scale
    cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-20)) if r.numel() >= 4096 else 1.0
    worst.append((err, 1.0 - cos, name))
    return err, cos


def print_report(tag, worst):
    import os
    if os.environ.get("PEVAE_TOL_REPORT"):
        worst.sort(reverse=True)
        print("\n[%s] worst relative gradient errors:" % tag, [(round(e, 5), n) for e, _, n in worst[:6]])
        print("[%s] worst 1 - cos:" % tag, sorted([(round(c, 7), n) for _, c, n in worst], reverse=True)[:4])


def _pair(cfg, dev):
    from pevae_amd.dls_vae import DLSVAE
    torch.manual_seed(0)
    gpu = DLSVAE(cfg)
    ref = RefDLSVAE(cfg)
    missing, unexpected = ref.load_state_dict(gpu.state_dict(), strict=False)
    assert not unexpected and not [m for m in missing if "unk_mask" not in m], (missing, unexpected)
    return gpu.to(dev), ref


def share_dropout_masks(gpu, ref, cfg, B, dev, seed=17):
    """The decoder's two dropouts (decoders.py:175 on the embedded inputs, :189 on the states; the encoder's input dropout
    has p = 0, dialog_models.py:530-534) draw the SAME supplied keep masks on both sides."""
    import oracle.ref_modules as R
    if cfg.dropout <= 0:
        return
    gen = torch.Generator().manual_seed(seed)
    L, E, H = cfg.max_seq_len, cfg.embed_size, cfg.dec_cell_size
    for name, shape in (("input_dropout", (B, L, E)), ("output_dropout", (B, L, H))):
        keep = torch.rand(shape, generator=gen) >= cfg.dropout
        getattr(gpu.x_decoder, name).mask = keep.to(dev)
        setattr(ref.x_decoder, name, R.MaskedDropout(keep, cfg.dropout))


@pytest.mark.parametrize("kw", [dict(), dict(use_attention=True), dict(do_pred=True), dict(latent_stack="dls"),
                                dict(dropout=0.3), dict(dropout=0.3, use_attention=True),
                                dict(dropout=0.3, latent_stack="dls"),
                                dict(latent_stack="dls", use_mutual=True, do_pred=True),        # configs[4] wiring
                                dict(use_feature=True, n_feature=13),
                                dict(latent_stack="dls", use_mutual=True, use_feature=True, n_feature=5, dropout=0.3)])
def test_step_matches_oracle(dev, kw):
    from pevae_amd import ops
    from pevae_amd.dls_vae import default_config, synthetic_batch
    from pevae_amd.optim import FlatAdam
    kw = dict(dict(dropout=0.0), **kw)
    cfg = default_config(vocab_size=517, embed_size=20, dec_cell_size=64, max_seq_len=11, n_user=40, n_item=30,
                         latent_size=16, batch_size=24, **kw)
    gpu, ref = _pair(cfg, dev)
    state0 = {k: v.detach().cpu().clone() for k, v in gpu.state_dict().items()}    # (spectral-norm u / v advance in a training forward)
    gpu.train()
    ref.train()
    share_dropout_masks(gpu, ref, cfg, 24, dev)     # p = 0.3: the bench's arithmetic, with the masks fed to the oracle
    feed = synthetic_batch(cfg, 24, 99)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    gen = torch.Generator().manual_seed(3)
    if kw.get("use_mutual"):                   # DeNAuto: uniform draws of its two levels + Gumbel noise of the hard negatives
        eps = ([torch.rand(24, d, generator=gen) for d in (4, 8)],
               [-torch.empty(24, 24).exponential_(generator=gen).log() for _ in range(2)])
    elif kw.get("latent_stack") == "dls":      # uniform draws of the three DLS_Auto levels (H/16, H/8, H/4 wide)
        eps = [torch.rand(24, d, generator=gen) for d in (4, 8, 16)]
    else:
        eps = torch.randn(24, 16, generator=gen)

    def to_dev(e):
        if isinstance(e, (list, tuple)):
            return type(e)(to_dev(t) for t in e)
        return e.to(dev)

    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-3)
    loss_r = ref(feed, eps=eps)
    ref.valid_loss(loss_r).backward()
    # (parameters the loss does not reach -- the last DeNAuto `layers` Block feeds nothing -- have no gradient: zeros)
    ref_grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in ref.named_parameters()}
    norm_r = torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0, 2)
    opt_r.step()

    opt = FlatAdam(gpu.parameters(), lr=1e-3, max_norm=5.0, guards=False)     # grads now sink into the flat buffer
    opt.zero_grad()
    loss = gpu(feed_d, eps=to_dev(eps))
    gpu.valid_loss(loss).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    # tolerances: bf16 MFMA operands in the GRU / vocabulary GEMMs (fp32 accumulate) vs the fp32 oracle
    assert abs(float(loss.vae_nll.detach()) - float(loss_r.vae_nll.detach())) <= 2e-2 * abs(float(loss_r.vae_nll.detach()))
    assert abs(float(loss.vae_PPL) - float(loss_r.vae_PPL)) <= 3e-2 * abs(float(loss_r.vae_PPL))
    assert abs(float(loss.vae_rkl) - float(loss_r.vae_rkl)) <= 2e-2 * abs(float(loss_r.vae_rkl)) + 1e-4
    for n, p in gpu.named_parameters():
        g, r = p.grad.cpu(), ref_grads[n]
        scale = float(r.abs().max()) + 1e-6
        assert float((g - r).abs().max()) <= 6e-2 * scale, (n, float((g - r).abs().max()), scale)
        # attention.dec_w only sees sum_j dscore[b,i,j] * (1 - tanh^2), and the softmax Jacobian makes
        # sum_j dscore = 0: its gradient is a second-order cancellation residue, so the ~1e-2 relative bf16 noise
        # of the GRU outputs feeding it gives O(1) relative error.  It keeps the max-abs bound above; the
        # attention kernels themselves are checked at fp32 tolerance in test_gpu_modules.py::test_attention.
        if r.numel() > 64 and "attention.dec_w" not in n:
            cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-12))
            assert cos > 0.995, (n, cos)
    # The same step against the oracle with the matrix core's operand rounding emulated (what the full-size tests use): the
    # bf16 format no longer hides anything, so every parameter -- attention.dec_w included -- is held to 1e-2 of its largest
    # entry, and every parameter but that one to a cosine of 0.9995.  (attention.dec_w at THIS size stays out of the cosine
    # check only: sum_j dscore[b,i,j] = 0 makes its gradient a cancellation residue whose direction is set by the last bits
    # of the recurrence's states -- measured cosine 0.83 against the emulating oracle with every entry within 1e-2 of the
    # largest; at full size, test_step_full_size_attention_oracle, it passes the 1 - cos <= 5e-5 bound like everything else.)
    ref2 = RefDLSVAE(cfg)
    ref2.load_state_dict(state0, strict=False)
    ref2.train()
    share_dropout_masks(gpu, ref2, cfg, 24, dev)
    with bf16_oracle():
        loss_b = ref2(feed, eps=eps)
        ref2.valid_loss(loss_b).backward()
    assert abs(float(loss.vae_nll.detach()) - float(loss_b.vae_nll.detach())) <= 4e-3 * abs(float(loss_b.vae_nll.detach()))
    assert abs(float(loss.vae_rkl) - float(loss_b.vae_rkl)) <= 4e-3 * abs(float(loss_b.vae_rkl)) + 1e-4
    for n, p in ref2.named_parameters():
        r = p.grad if p.grad is not None else torch.zeros_like(p)
        g = dict(gpu.named_parameters())[n].grad.cpu()
        scale = float(r.abs().max()) + 1e-6
        assert float((g - r).abs().max()) <= 1e-2 * scale, ("bf16 oracle", n, float((g - r).abs().max()), scale)
        if r.numel() > 64 and "attention.dec_w" not in n:
            cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-12))
            assert cos > 0.9995, ("bf16 oracle", n, cos)
    opt.step()
    torch.cuda.synchronize()
    st = opt.state_dict_host()
    assert abs(st["last_norm"] - float(norm_r)) <= 3e-2 * float(norm_r)
    ref_params = dict(ref.named_parameters())
    for n, p in gpu.named_parameters():
        d = float((p.detach().cpu() - ref_params[n].detach()).abs().max())
        assert d <= 2.5e-3, (n, d)        # one Adam step moves every entry by <= lr = 1e-3 (sign of a noisy grad may flip)


def test_step_eval_and_second_backward_accumulates(dev):
    from pevae_amd.dls_vae import default_config, synthetic_batch
    cfg = default_config(vocab_size=300, embed_size=12, dec_cell_size=32, max_seq_len=7, n_user=10, n_item=10,
                         latent_size=8, dropout=0.0, batch_size=8)
    gpu, ref = _pair(cfg, dev)
    feed = synthetic_batch(cfg, 8, 5)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    gpu.eval()
    ref.eval()
    with torch.no_grad():
        a, b = gpu(feed_d), ref(feed)
    assert a.vae_rkl is None and b.vae_rkl is None
    assert abs(float(a.vae_nll) - float(b.vae_nll)) <= 2e-2 * abs(float(b.vae_nll))
    # two backward passes without zeroing: the in-place gradient sinks must accumulate like autograd does
    gpu.train()
    eps = torch.randn(8, 8, generator=torch.Generator().manual_seed(1)).to(dev)
    gpu.valid_loss(gpu(feed_d, eps=eps)).backward()
    g1 = {n: p.grad.clone() for n, p in gpu.named_parameters()}
    gpu.valid_loss(gpu(feed_d, eps=eps)).backward()
    for n, p in gpu.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * g1[n].cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("kw", [dict(latent_stack="dls", dropout=0.3), dict(do_pred=True),
                                dict(latent_stack="dls", use_mutual=True, do_pred=True), dict(use_attention=True)])
def test_split_backward_schedules_give_the_unsplit_gradients(dev, kw):
    """The data-parallel overlap schedule (bench.py, N > 1): DLSVAE.backward_vocab / backward_mid / backward_encoder with
    the autograd graph cut behind the vocabulary projection (split_backward = 1) and also in front of the utterance
    encoder (= 2; with attention the second cut is not made) leave bit-identical gradients to the one-pass backward, and
    after backward_mid every gradient but the encoder's and the word table's is already final."""
    from pevae_amd import ops
    from pevae_amd.dls_vae import DLSVAE, default_config, synthetic_batch
    from pevae_amd.optim import FlatAdam
    cfg = default_config(vocab_size=517, embed_size=20, dec_cell_size=64, max_seq_len=11, n_user=40, n_item=30,
                         latent_size=16, batch_size=24, **dict(dict(dropout=0.0), **kw))
    torch.manual_seed(0)
    gpu = DLSVAE(cfg).to(dev)
    gpu.train()
    opt = FlatAdam(gpu.parameters(), lr=1e-3, max_norm=5.0, guards=False)
    feed = {k: v.to(dev) for k, v in synthetic_batch(cfg, 24, 99).items()}
    gen = torch.Generator().manual_seed(3)
    if kw.get("use_mutual"):
        eps = ([torch.rand(24, d, generator=gen).to(dev) for d in (4, 8)],
               [(-torch.empty(24, 24).exponential_(generator=gen).log()).to(dev) for _ in range(2)])
    elif kw.get("latent_stack") == "dls":
        eps = [torch.rand(24, d, generator=gen).to(dev) for d in (4, 8, 16)]
    else:
        eps = torch.randn(24, 16, generator=gen).to(dev)
    if cfg.dropout > 0:                        # the same keep masks in every pass
        for name, shape in (("input_dropout", (24, 11, 20)), ("output_dropout", (24, 11, 64))):
            getattr(gpu.x_decoder, name).mask = (torch.rand(shape, generator=gen) >= cfg.dropout).to(dev)
    state = {n: b.clone() for n, b in gpu.named_buffers()}        # spectral-norm u / v advance in every training forward

    def run(split):
        for n, b in gpu.named_buffers():
            b.copy_(state[n])
        gpu.split_backward = split
        opt.zero_grad()
        loss = gpu(feed, eps=eps)
        mid = None
        if split:
            gpu.backward_vocab(loss)
            gpu.backward_mid(loss)
            mid = opt.flat_g.clone()
            gpu.backward_encoder(loss)
        else:
            gpu.backward(0, loss)
        torch.cuda.synchronize()
        ops.gru_check_all()
        return opt.flat_g.clone(), mid

    want, _ = run(False)
    assert float(want.abs().sum()) > 0
    late = set(id(p) for p in [gpu.x_embedding.weight] + list(gpu.x_encoder.parameters()))
    for split in (1, 2):
        got, mid = run(split)
        assert torch.equal(got, want), split
        if split == 2 and not cfg.use_attention:
            assert gpu._cut_enc is None
            for p in opt.params:
                a, b = opt.param_range(p)
                if id(p) not in late:
                    assert torch.equal(mid[a:b], want[a:b])
            a, b = opt.param_range(gpu.x_encoder.rnn.layer_weights(0, 0)[1])
            assert float(mid[a:b].abs().sum()) == 0.0                # the encoder has not run backward yet
    gpu.split_backward = False


def test_schedule_switches_leave_the_gradients_bitwise_alone(dev):
    """Round-3 schedule changes move work, they do not change it: the decoder's teacher-forced input (embedding, dropout,
    input projection) and the lookups' id sorts made beside the latent stack, the backward pass's weight transposes made
    during the forward pass, the zero-grad fill beside the head of the forward pass, the user / item / rating
    lookups beside the encoder's forward recurrence -- each alone and all together leave
    every gradient bit-identical to the step with all of them off; the weight-gradient GEMMs over K-major operands
    (no transposed copies) agree with the copies' path to fp32 accumulation order."""
    from pevae_amd import dls_vae as dv, ops, optim
    from pevae_amd.dls_vae import DLSVAE, default_config, synthetic_batch
    from pevae_amd.optim import FlatAdam
    B, Lq = 32, 10                                          # 10 x 32 time-major rows: a multiple of 64 (the tn GEMM's K)
    cfg = default_config(vocab_size=517, embed_size=24, dec_cell_size=64, max_seq_len=Lq, n_user=40, n_item=30,
                         latent_size=16, batch_size=B, latent_stack="dls", dropout=0.3)
    torch.manual_seed(0)
    gpu = DLSVAE(cfg).to(dev)
    gpu.train()
    opt = FlatAdam(gpu.parameters(), lr=1e-3, max_norm=5.0, guards=False)
    feed = {k: v.to(dev) for k, v in synthetic_batch(cfg, B, 99).items()}
    gen = torch.Generator().manual_seed(3)
    eps = [torch.rand(B, d, generator=gen).to(dev) for d in (4, 8, 16)]
    for name, shape in (("input_dropout", (B, Lq, 24)), ("output_dropout", (B, Lq, 64))):
        getattr(gpu.x_decoder, name).mask = (torch.rand(shape, generator=gen) >= cfg.dropout).to(dev)
    state = {n: b.clone() for n, b in gpu.named_buffers()}
    names = ("HOIST_DECODER_INPUT", "PRESORT_SIDE", "PRETRANSPOSE", "GEMM_TN")
    saved = {n: getattr(ops, n) for n in names}
    saved_zero, saved_head = optim.ZERO_BESIDE, dv.HEAD_SIDE

    def run(**on):
        dv.HEAD_SIDE = bool(on.get("HEAD_SIDE", False))
        for n, b in gpu.named_buffers():
            b.copy_(state[n])
        for n in names:
            setattr(ops, n, bool(on.get(n, False)))
        optim.ZERO_BESIDE = bool(on.get("ZERO_BESIDE", False))
        opt.zero_grad(beside_forward=True)
        loss = gpu(feed, eps=eps)
        gpu.backward(0, loss)
        torch.cuda.synchronize()
        ops.gru_check_all()
        return opt.flat_g.clone(), float(loss.vae_nll.detach())

    try:
        want, nll = run()
        assert float(want.abs().sum()) > 0
        for sw in ("HOIST_DECODER_INPUT", "PRESORT_SIDE", "PRETRANSPOSE", "ZERO_BESIDE"):
            got, nll2 = run(**{sw: True})
            assert nll2 == nll and torch.equal(got, want), sw
        got, nll2 = run(HOIST_DECODER_INPUT=True, PRESORT_SIDE=True, PRETRANSPOSE=True, ZERO_BESIDE=True)
        assert nll2 == nll and torch.equal(got, want)
        # the user / item / rating lookups beside the encoder's recurrence
        got, nll2 = run(HEAD_SIDE=True)
        assert nll2 == nll and torch.equal(got, want), "HEAD_SIDE"
        got, nll2 = run(HOIST_DECODER_INPUT=True, PRESORT_SIDE=True, PRETRANSPOSE=True, ZERO_BESIDE=True, HEAD_SIDE=True)
        assert nll2 == nll and torch.equal(got, want), "all"
        got, nll2 = run(GEMM_TN=True)
        assert nll2 == nll
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6 * float(want.abs().max()))
    finally:
        for n in names:
            setattr(ops, n, saved[n])
        optim.ZERO_BESIDE = saved_zero
        dv.HEAD_SIDE = saved_head


def test_step_full_size_oracle_and_properties(dev):
    """BASELINE.json configs[1] at full size (bs=256, L=50, V=20004, H=512, DLS pyramid): the composed step against the
    CPU oracle (a few seconds on the host cores), plus properties that hold at any size."""
    from pevae_amd import ops
    from pevae_amd.dls_vae import PAD_ID, default_config, synthetic_batch
    cfg = default_config(dropout=0.0, batch_size=256, latent_stack="dls")
    gpu, ref = _pair(cfg, dev)
    gpu.train()
    ref.train()
    feed = synthetic_batch(cfg, 256, 4321)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    H = cfg.dec_cell_size
    gen = torch.Generator().manual_seed(11)
    eps = [torch.rand(256, d, generator=gen) for d in (H // 16, H // 8, H // 4)]
    eps_d = [e.to(dev) for e in eps]

    from pevae_amd.optim import FlatAdam
    opt = FlatAdam(gpu.parameters(), lr=1e-4, max_norm=5.0, guards=False)     # gradients sink into the flat buffer, as in bench.py

    buffers = {n: b.clone() for n, b in gpu.named_buffers()}      # spectral-norm u vectors advance on every training forward

    def run(fd):
        with torch.no_grad():
            for n, b in gpu.named_buffers():
                b.copy_(buffers[n])
        opt.zero_grad()
        loss = gpu(fd, eps=eps_d)
        gpu.valid_loss(loss).backward()
        torch.cuda.synchronize()
        ops.gru_check_all()
        return loss, {n: p.grad.clone() for n, p in gpu.named_parameters()}

    loss, grads = run(feed_d)

    # (1) the oracle at full size, its matrix-core products rounded to bf16 like the kernels' (forward and backward)
    with bf16_oracle():
        loss_r = ref(feed, eps=eps)
        ref.valid_loss(loss_r).backward()
    assert abs(float(loss.vae_nll.detach()) - float(loss_r.vae_nll.detach())) <= FULL_LOSS_RTOL * abs(float(loss_r.vae_nll.detach()))
    assert abs(float(loss.vae_rkl.detach()) - float(loss_r.vae_rkl.detach())) <= FULL_LOSS_RTOL * abs(float(loss_r.vae_rkl.detach())) + 1e-4
    worst = []
    for n, p in ref.named_parameters():
        err, cos = grad_report(n, grads[n].cpu(), p.grad, worst)
    print_report("yelp", worst)
    for err, omc, n in worst:
        assert err <= FULL_GRAD_TOL, (n, err)
        assert omc <= FULL_COS_TOL, (n, omc)

    # (2) no atomics anywhere on the path: a second run reproduces every bit
    loss2, grads2 = run(feed_d)
    assert float(loss.vae_nll) == float(loss2.vae_nll) and float(loss.vae_rkl) == float(loss2.vae_rkl)
    for n in grads:
        assert torch.equal(grads[n], grads2[n]), n

    # (3) softmax rows sum to one: the vocabulary bias gradient sums to zero over V
    db = grads["x_decoder.project.bias"].double()
    assert abs(float(db.sum())) <= 1e-3 * float(db.abs().sum())

    # (4) rows nobody looked up carry exactly zero gradient (sorted segmented sum, no scatter over the table);
    #     padding_idx row of the word table: nn.Embedding(padding_idx=0)
    gw = grads["x_embedding.weight"]
    used = torch.zeros(cfg.vocab_size, dtype=torch.bool, device=dev)
    used[feed_d["sentence_embeddings"].reshape(-1)] = True
    used[feed_d["input_ids"].reshape(-1)] = True
    used[PAD_ID] = False
    assert not gw[~used].any() and gw[used].abs().sum(1).gt(0).float().mean() > 0.99
    for tab, key in (("user_embedding.weight", "user_ids"), ("item_embedding.weight", "item_ids")):
        hit = torch.zeros(grads[tab].shape[0], dtype=torch.bool, device=dev)
        hit[feed_d[key]] = True
        assert not grads[tab][~hit].any(), tab

    # (5) ignored targets: decoder inputs behind the last live target cannot move the loss or the vocabulary gradients'
    #     live rows (NLLLoss(ignore_index=0), criterions.py:38-41) -- bit for bit, since the live rows are compacted
    #     in the same order
    fd = dict(feed_d)
    ids = feed_d["input_ids"].clone()
    live = feed_d["ans_ids"] != PAD_ID
    last = (live.long() * torch.arange(ids.shape[1], device=dev)).max(1).values
    behind = torch.arange(ids.shape[1], device=dev).unsqueeze(0) > last.unsqueeze(1)
    ids[behind] = 7
    fd["input_ids"] = ids
    loss3, grads3 = run(fd)
    assert float(loss3.vae_nll) == float(loss.vae_nll)
    assert torch.equal(grads3["x_decoder.project.weight"], grads["x_decoder.project.weight"])


def test_step_full_size_dropout_oracle(dev):
    """BASELINE.json configs[1] at full size WITH the bench's dropout (p = 0.3 on the decoder's embedded inputs and on its
    states, decoders.py:175,189): the keep masks are supplied to both sides, the oracle emulates the matrix core's operand
    rounding, and the bounds are the full-size ones (4e-3 of the largest entry, 1 - cos <= 5e-5)."""
    from pevae_amd.dls_vae import default_config, synthetic_batch
    cfg = default_config(dropout=0.3, batch_size=256, latent_stack="dls")
    gpu, ref = _pair(cfg, dev)
    gpu.train()
    ref.train()
    share_dropout_masks(gpu, ref, cfg, 256, dev, seed=23)
    feed = synthetic_batch(cfg, 256, 4322)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    H = cfg.dec_cell_size
    gen = torch.Generator().manual_seed(12)
    eps = [torch.rand(256, d, generator=gen) for d in (H // 16, H // 8, H // 4)]
    loss, grads = _full_size_vs_oracle(gpu, ref, feed, feed_d, eps, [e.to(dev) for e in eps])
    # dropout really was on: a third of the decoder states' columns are dead in every row, so the projection's gradient
    # differs from the one without dropout (the p = 0 test above runs the same seed family)
    assert float(loss.vae_nll) > 0 and grads["x_decoder.project.weight"].abs().sum() > 0


def test_step_full_size_attention_oracle(dev):
    """BASELINE.json configs[2] at full size: flat Gaussian latent + decoder attention over the encoder states
    (bs=256, L=50, H=512; the oracle materialises what the fused attention kernels never store)."""
    from pevae_amd import ops
    from pevae_amd.dls_vae import default_config, synthetic_batch
    cfg = default_config(dropout=0.0, batch_size=256, latent_stack="gauss", use_attention=True)
    gpu, ref = _pair(cfg, dev)
    gpu.train()
    ref.train()
    feed = synthetic_batch(cfg, 256, 777)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    eps = torch.randn(256, cfg.latent_size, generator=torch.Generator().manual_seed(5))

    def run():
        gpu.zero_grad(set_to_none=True)
        loss = gpu(feed_d, eps=eps.to(dev))
        gpu.valid_loss(loss).backward()
        torch.cuda.synchronize()
        ops.gru_check_all()
        return loss, {n: p.grad.clone() for n, p in gpu.named_parameters()}

    loss, grads = run()
    with bf16_oracle():
        loss_r = ref(feed, eps=eps)
        ref.valid_loss(loss_r).backward()
    assert abs(float(loss.vae_nll.detach()) - float(loss_r.vae_nll.detach())) <= FULL_LOSS_RTOL * abs(float(loss_r.vae_nll.detach()))
    assert abs(float(loss.vae_rkl.detach()) - float(loss_r.vae_rkl.detach())) <= FULL_LOSS_RTOL * abs(float(loss_r.vae_rkl.detach())) + 1e-4
    worst = []
    for n, p in ref.named_parameters():
        if "query_w.bias" in n:          # softmax shift invariance: exactly zero in exact arithmetic, rounding on both sides
            assert float(grads[n].abs().max()) <= 1e-6 + 1e-3 * float(p.grad.abs().max())
            continue
        grad_report(n, grads[n].cpu(), p.grad, worst)
    print_report("attention", worst)
    for err, omc, n in worst:
        assert err <= FULL_GRAD_TOL, (n, err)
        assert omc <= FULL_COS_TOL, (n, omc)
    loss2, grads2 = run()
    assert float(loss.vae_nll.detach()) == float(loss2.vae_nll.detach())
    for n in grads:
        assert torch.equal(grads[n], grads2[n]), n


@pytest.mark.parametrize("stack", ["gauss", "dls"])
def test_step_golden_from_reference_modules(dev, stack):
    """pevae_amd.dls_vae.DLSVAE against the step composed from the reference's own EncoderRNN / DecoderRNN /
    Hidden2Gaussian / LinearConnector / NLLEntropy / NormKLLoss (and DlsRec / DlsPri / DLS_Auto / Block) classes
    (tests/golden/composed_step_*.npz, oracle/make_golden.py gen_composed_step): losses and every parameter gradient.
    Tolerances: bf16 MFMA operands in the recurrences and the vocabulary GEMMs, fp32 accumulate."""
    from conftest import load_golden, load_params
    from pevae_amd import ops
    from pevae_amd.dls_vae import DLSVAE
    from test_oracle_golden import composed_step_inputs
    g = load_golden("composed_step_" + stack)
    cfg, feed, eps = composed_step_inputs(g, lambda a: torch.from_numpy(np.array(a)).to(dev))
    m = load_params(DLSVAE(cfg), g).to(dev)
    m.train()
    loss = m(feed, eps=eps)
    m.valid_loss(loss).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    for got, key in ((loss.vae_nll, "vae_nll"), (loss.vae_PPL, "vae_ppl"), (loss.vae_rkl, "vae_rkl")):
        assert abs(float(got.detach()) - float(g[key])) <= 2e-2 * abs(float(g[key])) + 1e-4, key
    n = 0
    for k, p in m.named_parameters():
        if "g." + k in g:
            r = torch.from_numpy(g["g." + k])
            d = float((p.grad.cpu() - r).abs().max())
            assert d <= 6e-2 * float(r.abs().max()) + 1e-7, (k, d, float(r.abs().max()))
            n += 1
    assert n == len([k for k in g if k.startswith("g.")])


def _full_size_vs_oracle(gpu, ref, feed, feed_d, eps, eps_d, skip_cos=()):
    from pevae_amd import ops
    gpu.zero_grad(set_to_none=True)
    loss = gpu(feed_d, eps=eps_d)
    gpu.valid_loss(loss).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    with bf16_oracle():
        loss_r = ref(feed, eps=eps)
        ref.valid_loss(loss_r).backward()
    for key in ("vae_nll", "vae_rkl", "rating_mse"):
        if loss_r.get(key) is not None:
            a, b = float(loss[key].detach()), float(loss_r[key].detach())
            assert abs(a - b) <= FULL_LOSS_RTOL * abs(b) + 1e-4, (key, a, b)
    grads = {n: p.grad for n, p in gpu.named_parameters()}
    worst = []
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert grads[n] is None or not grads[n].any(), n
            continue
        grad_report(n, grads[n].cpu(), p.grad, worst)
    print_report("full", worst)
    for err, omc, n in worst:
        assert err <= FULL_GRAD_TOL, (n, err)
        if not any(s in n for s in skip_cos):
            assert omc <= FULL_COS_TOL, (n, omc)
    return loss, grads


def test_step_full_size_mutual_oracle(dev):
    """BASELINE.json configs[4], the per-GPU shard (bs=256 of the global 2048): --use_mutual (DEM discriminator stack,
    DeNAuto) + --do_pred rating head at full size against the oracle, with the uniform draws and the Gumbel noise of the
    hard negatives supplied to both."""
    from pevae_amd.dls_vae import default_config, synthetic_batch
    cfg = default_config(dropout=0.0, batch_size=256, latent_stack="dls", use_mutual=True, do_pred=True)
    gpu, ref = _pair(cfg, dev)
    gpu.train()
    ref.train()
    feed = synthetic_batch(cfg, 256, 2048)
    feed_d = {k: v.to(dev) for k, v in feed.items()}
    H = cfg.dec_cell_size
    gen = torch.Generator().manual_seed(13)
    eps = ([torch.rand(256, d, generator=gen) for d in (H // 16, H // 8)],
           [-torch.empty(256, 256).exponential_(generator=gen).log() for _ in range(2)])
    eps_d = tuple([t.to(dev) for t in e] for e in eps)
    loss, _ = _full_size_vs_oracle(gpu, ref, feed, feed_d, eps, eps_d)
    assert loss.get("rating_mse") is not None and float(loss.vae_rkl) > 0


def test_step_full_size_book_tables_and_sparse_exchange(dev):
    """BASELINE.json configs[3], Amazon-Book-shape tables (200 000 users x 256, 100 000 items x 256 = 77 M of the 107 M
    parameters) with the DLS latent stack: the step against the oracle, exact zeros in every row nobody looked up, and the data-parallel exchange
    of those tables at this size -- two ranks' (ids, rows) contributions (dp.pack_sparse_rows) merged on the device
    (dp.merge_sparse_rows, the library's sorted segmented sum) equal the dense sum of the two table gradients bit for bit
    on both ranks."""
    from pevae_amd import dp
    from pevae_amd.dls_vae import default_config, synthetic_batch
    # latent_stack="dls": the stack `bench.py --config book` times (round 2 checked the flat one here)
    cfg = default_config(dropout=0.0, batch_size=256, latent_stack="dls", n_user=200000, n_item=100000)
    gpu, ref = _pair(cfg, dev)
    gpu.train()
    ref.train()
    feeds = [synthetic_batch(cfg, 256, 3000 + r) for r in range(2)]
    H = cfg.dec_cell_size
    gen = torch.Generator().manual_seed(5)
    eps = [torch.rand(256, d, generator=gen) for d in (H // 16, H // 8, H // 4)]
    eps_d = [e.to(dev) for e in eps]
    table_grads = []
    for r, feed in enumerate(feeds):
        feed_d = {k: v.to(dev) for k, v in feed.items()}
        if r == 0:
            _, grads = _full_size_vs_oracle(gpu, ref, feed, feed_d, eps, eps_d)
        else:
            gpu.zero_grad(set_to_none=True)
            gpu.valid_loss(gpu(feed_d, eps=eps_d)).backward()
            grads = {n: p.grad for n, p in gpu.named_parameters()}
        per = {}
        for tab, key in (("user_embedding.weight", "user_ids"), ("item_embedding.weight", "item_ids")):
            gt = grads[tab]
            hit = torch.zeros(gt.shape[0], dtype=torch.bool, device=dev)
            hit[feed_d[key]] = True
            assert not gt[~hit].any() and bool(gt[hit].abs().sum(1).gt(0).all()), tab
            per[tab] = (gt.clone(), feed_d[key])
        table_grads.append(per)
    for tab in ("user_embedding.weight", "item_embedding.weight"):
        packed = [dp.pack_sparse_rows(*table_grads[r][tab]) for r in range(2)]
        ids_all = torch.cat([p[0] for p in packed])
        rows_all = torch.cat([p[1] for p in packed])
        want = table_grads[0][tab][0] + table_grads[1][tab][0]
        merged = []
        for r in range(2):
            mine = table_grads[r][tab][0].clone()
            dp.merge_sparse_rows(mine, table_grads[r][tab][1], ids_all, rows_all)
            torch.testing.assert_close(mine, want, rtol=1e-6, atol=1e-7)
            merged.append(mine)
        assert torch.equal(merged[0], merged[1])
