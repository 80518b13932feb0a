"""Pins the CPU oracle (oracle/ref_modules.py) against every golden vector generated from the
reference's own modules (oracle/make_golden.py -> tests/golden/*.npz), plus known-answer identities
and stock torch kernels as an independent second opinion."""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_params
from oracle import ref_modules as R

TOL = dict(rtol=2e-5, atol=2e-6)     # fp32 vs fp32, different summation order


def T(a):
    return torch.from_numpy(np.array(a))


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), **tol)


def check_grads(module, g, prefix="g.", **kw):
    n = 0
    for k, p in module.named_parameters():
        if prefix + k in g:
            close(p.grad, g[prefix + k], **kw)
            n += 1
    assert n > 0


@pytest.mark.parametrize("tag,kw", [("gru", {}), ("gru_mid", {}), ("gru_bi", dict(bidirection=True)),
                                    ("gru_2l", dict(n_layers=2)), ("lstm", dict(rnn_cell="lstm"))])
def test_encoder(tag, kw):
    g = load_golden("encoder_" + tag)
    B, L, E = g["x"].shape
    H = g["h0"].shape[-1]
    enc = load_params(R.EncoderRNN(E, H, **kw), g)
    x = T(g["x"]).requires_grad_()
    if "c0" in g:
        out, (hn, cn) = enc(x, init_state=(T(g["h0"]), T(g["c0"])))
        loss = (out * T(g["w_out"])).sum() + (hn * T(g["w_hid"])).sum() + (cn * T(g["w_hid"])).sum() * 0.5
        close(cn, g["cn"])
    else:
        out, hn = enc(x, init_state=T(g["h0"]))
        loss = (out * T(g["w_out"])).sum() + (hn * T(g["w_hid"])).sum()
    loss.backward()
    close(out, g["out"])
    close(hn, g["hn"])
    close(x.grad, g["dx"], rtol=1e-4, atol=1e-5)
    check_grads(enc, g, rtol=1e-4, atol=1e-5)
    out0, hn0 = enc(T(g["x"]))
    close(out0, g["out_nostate"])
    close(hn0[0] if isinstance(hn0, tuple) else hn0, g["hn_nostate"])


def _utt_encoder_case(cls, g, tag, to=lambda a: T(a)):
    kw = dict(bidirection=True, use_attn=True, feat_size=1) if tag == "attn" else dict(bidirection=False, use_attn=False)
    enc = load_params(cls(16, 0.0, rnn_cell="gru", vocab_size=23, embed_dim=8, **kw), g)
    feats = to(g["feats"]) if "feats" in g else None
    init = to(g["init_state"]) if "init_state" in g else None
    return enc, feats, init


@pytest.mark.parametrize("tag", ["attn", "last"])
def test_utt_encoder(tag):
    """oracle RnnUttEncoder vs the reference class (encoders.py:69-135): utterance vectors, states, attention weights and
    every parameter gradient."""
    g = load_golden("utt_encoder_" + tag)
    enc, feats, init = _utt_encoder_case(R.RnnUttEncoder, g, tag)
    utt, enc_outs, enc_last, attn = enc(T(g["utterances"]), feats=feats, init_state=init, return_all=True)
    (utt * T(g["w"])).sum().backward()
    close(utt, g["utt"])
    close(enc_outs, g["enc_outs"])
    close(enc_last, g["enc_last"])
    if tag == "attn":
        close(attn, g["attn"])
    check_grads(enc, g, rtol=1e-4, atol=1e-5)


def test_encoder_varlen():
    g = load_golden("encoder_gru_varlen")
    enc = load_params(R.EncoderRNN(8, 16, variable_lengths=True), g)
    out, hn = enc(T(g["x"]), input_lengths=g["lens"].tolist())
    close(out, g["out"])
    close(hn, g["hn"])


def test_encoder_bidirectional_varlen():
    g = load_golden("encoder_gru_bi_varlen")
    enc = load_params(R.EncoderRNN(8, 16, variable_lengths=True, bidirection=True), g)
    x = T(g["x"]).requires_grad_()
    out, hn = enc(x, input_lengths=g["lens"].tolist())
    ((out * T(g["w"])).sum() + hn.sum()).backward()
    close(out, g["out"])
    close(hn, g["hn"])
    close(x.grad, g["dx"])


@pytest.mark.parametrize("tag,bi", [("lstm_varlen", False), ("lstm_bi_varlen", True)])
def test_encoder_lstm_varlen(tag, bi):
    g = load_golden("encoder_" + tag)
    enc = load_params(R.EncoderRNN(8, 16, rnn_cell="lstm", variable_lengths=True, bidirection=bi), g)
    x = T(g["x"]).requires_grad_()
    out, (hn, cn) = enc(x, input_lengths=g["lens"].tolist())
    ((out * T(g["w"])).sum() + hn.sum() + 0.5 * cn.sum()).backward()
    close(out, g["out"])
    close(hn, g["hn"])
    close(cn, g["cn"])
    close(x.grad, g["dx"])


def test_gru_cell_matches_torch_gru():
    torch.manual_seed(0)
    gru = torch.nn.GRU(10, 16, batch_first=True)
    enc = R.EncoderRNN(10, 16)
    enc.rnn.load_state_dict(gru.state_dict())
    x = torch.randn(5, 9, 10)
    o1, h1 = gru(x)
    o2, h2 = enc(x)
    close(o2, o1.detach())
    close(h2, h1.detach())


@pytest.mark.parametrize("tag,kw,lat", [("gru", dict(rnn_cell="gru"), 0), ("gru_latent", dict(rnn_cell="gru"), 5),
                                        ("gru_temp", dict(rnn_cell="gru", softmax_temperature=0.7), 0),
                                        ("lstm", dict(rnn_cell="lstm"), 0)])
def test_decoder_teacher_forcing(tag, kw, lat):
    g = load_golden("decoder_tf_" + tag)
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(R.DecoderRNN(V, L, E + lat, H, 1, 2, 3, use_gpu=False, embedding=emb, **kw), g)
    h0 = T(g["h0"]).requires_grad_()
    init = h0
    if "c0" in g:
        c0 = T(g["c0"]).requires_grad_()
        init = (h0, c0)
    latent = T(g["latent"]) if lat else None
    logp, hid, ret = dec(B, T(g["ids"]), init, mode=R.TEACH_FORCE, latent_variable=latent)
    (logp * T(g["w"])).sum().backward()
    close(logp, g["logp"], rtol=1e-4, atol=1e-5)
    close(hid[0] if isinstance(hid, tuple) else hid, g["hid"])
    close(h0.grad, g["dh0"], rtol=1e-4, atol=1e-5)
    check_grads(dec, g, rtol=2e-4, atol=2e-5)
    assert ret["length"] == g["length"].tolist()
    assert len(ret["sequence"]) == int(g["n_seq"]) == 0      # decoders.py:239-242,286-292


@pytest.mark.parametrize("tag,mask_unk", [("greedy", False), ("greedy_maskunk", True)])
def test_decoder_gen_greedy(tag, mask_unk):
    g = load_golden("decoder_gen_" + tag)
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(R.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", use_gpu=False, embedding=emb,
                                   mask_unk=mask_unk), g)
    dec.eval()
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"]), T(g["h0"]), mode=R.GEN, gen_type="greedy")
    seq = torch.cat(ret["sequence"], dim=1)
    assert torch.equal(seq, T(g["seq"]))                     # symbols are bit-exact
    assert ret["length"] == g["length"].tolist()
    close(logp, g["logp"], rtol=1e-4, atol=1e-5)
    close(hid, g["hid"])


@pytest.mark.parametrize("gen_type", ["sample", "beam"])
def test_decoder_gen_sample_beam(gen_type):
    """decoders.py:255-270, :307-328.  `sample` golden: the reference's own run, with torch.multinomial's Exp(1) draws
    stored as input; `beam` golden: the reference's forward_step per token under its beam rule with floor division (the
    reference's own back-tracking raises on this torch; oracle/make_golden.py gen_decoder_sample_beam)."""
    g = load_golden("decoder_gen_" + gen_type)
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(R.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", use_gpu=False, embedding=emb), g)
    dec.eval()
    if gen_type == "sample":
        dec.sample_noise = T(g["q"])
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"]), T(g["h0"]), mode=R.GEN, gen_type=gen_type, beam_size=3)
    assert torch.equal(torch.cat(ret["sequence"], dim=1), T(g["seq"]))
    assert ret["length"] == g["length"].tolist()
    close(logp, g["logp"], rtol=1e-4, atol=1e-5)
    close(hid, g["hid"])


@pytest.mark.parametrize("mode", ["dot", "general", "cat"])
def test_attention(mode):
    g = load_golden("attention_" + mode)
    att = load_params(R.Attention(16, 16, mode, project=True), g)
    out, ctx = T(g["out"]).requires_grad_(), T(g["ctx"]).requires_grad_()
    y, a = att(out, ctx)
    (y * T(g["w"])).sum().backward()
    close(y, g["y"])
    close(a, g["attn"])
    close(out.grad, g["dout"], rtol=1e-4, atol=1e-5)
    close(ctx.grad, g["dctx"], rtol=1e-4, atol=1e-5)
    check_grads(att, g, rtol=1e-4, atol=1e-5)


def test_connectors():
    g = load_golden("hidden2gaussian")
    m = load_params(R.Hidden2Gaussian(16, 8), g)
    x = T(g["x"]).requires_grad_()
    mu, lv = m(x)
    ((mu * T(g["wm"])).sum() + (lv * T(g["wl"])).sum()).backward()
    close(mu, g["mu"]); close(lv, g["logvar"]); close(x.grad, g["dx"])
    check_grads(m, g)

    g = load_golden("hidden2gaussian_lstm")
    m = load_params(R.Hidden2Gaussian(16, 8, is_lstm=True), g)
    mu, lv = m((T(g["h"]), T(g["c"])))
    close(mu, g["mu"]); close(lv, g["logvar"])

    g = load_golden("linear_connector")
    m = load_params(R.LinearConnector(8, 16, False), g)
    z = T(g["z"]).requires_grad_()
    y = m(z)
    (y * T(g["w"])).sum().backward()
    close(y, g["y"]); close(z.grad, g["dz"]); check_grads(m, g)

    g = load_golden("linear_connector_lstm")
    m = load_params(R.LinearConnector(8, 16, True), g)
    yh, yc = m(T(g["z"]))
    close(yh, g["yh"]); close(yc, g["yc"])

    g = load_golden("greedy_connector")
    y, ids = R.GreedyConnector()(T(g["logits"]), False, return_max_id=True)
    assert torch.equal(ids, T(g["ids"])); close(y, g["y"])


@pytest.mark.parametrize("tag,kw", [("soft", {}), ("hard", dict(hard=True, return_max_id=True)),
                                    ("st", dict(straight_through=True))])
def test_gumbel(tag, kw):
    g = load_golden("gumbel_" + tag)
    logits = T(g["logits"]).requires_grad_()
    res = R.GumbelConnector()(logits, float(g["tau"]), False, u=T(g["u"]), **kw)
    y, ids = res if isinstance(res, tuple) else (res, None)
    (y * T(g["w"])).sum().backward()
    close(y, g["y"], rtol=1e-5, atol=1e-6)
    close(logits.grad, g["dlogits"], rtol=1e-4, atol=1e-6)
    if ids is not None:
        assert torch.equal(ids, T(g["ids"]))
    # same seed, no injected noise -> the reference's own CPU draw (nn_lib.py:231)
    torch.manual_seed(42)
    res2 = R.GumbelConnector()(T(g["logits"]), float(g["tau"]), False, **kw)
    close(res2[0] if isinstance(res2, tuple) else res2, g["y"], rtol=1e-5, atol=1e-6)


def test_criterions_nll():
    g = load_golden("criterions_nll")
    labels = T(g["labels"])
    for avg in ("seq", "word", None):
        logp = T(g["logp"]).requires_grad_()
        v = R.NLLEntropy(0, None, avg_type=avg)(logp, labels)
        v.backward()
        close(v, g["nll_%s" % avg])
        close(logp.grad, g["dlogp_%s" % avg])
    logp = T(g["logp"]).requires_grad_()
    v = R.Perplexity(0, None)(logp, labels)
    v.backward()
    close(v, g["ppl"]); close(logp.grad, g["dlogp_ppl"])
    # identities: PPL = exp(NLL_word); uniform log-probs => NLL_seq = n_tok * ln V / B
    close(torch.log(v), g["nll_word"], rtol=1e-6)
    B, L, V = g["logp"].shape
    uni = torch.full((B, L, V), -np.log(V), dtype=torch.float32)
    n_tok = int((labels != 0).sum())
    close(R.NLLEntropy(0, None)(uni, labels), n_tok * np.log(V) / B, rtol=1e-6)
    with pytest.raises(NotImplementedError):
        R.NLLEntropy(0, None, avg_type="bogus")


def test_criterions_normkl():
    g = load_golden("criterions_normkl")
    for ua in (False, True):
        ts = [T(g[k]).requires_grad_() for k in ("mq", "lq", "mp", "lp")]
        v = R.NormKLLoss(unit_average=ua)(ts[0], ts[1] * 0.5, ts[2], ts[3] * 0.5)
        v.backward()
        close(v, g["kl_ua%d" % ua])
        for n, t in zip(("mq", "lq", "mp", "lp"), ts):
            close(t.grad, g["d%s_ua%d" % (n, ua)], rtol=1e-4, atol=1e-6)
    assert abs(float(R.NormKLLoss()(T(g["mq"]), T(g["lq"]), T(g["mq"]), T(g["lq"])))) < 1e-6
    assert abs(float(g["kl_same"])) < 1e-6


def test_criterions_cat():
    g = load_golden("criterions_cat")
    lpy = T(g["log_py"])
    ck = R.CatKLLoss()
    for tag, kw in (("bs", dict(batch_size=4)), ("ua", dict(unit_average=True)), ("rows", dict(average=False))):
        lq = T(g["log_qy"]).requires_grad_()
        v = ck(lq, lpy, **kw)
        v.sum().backward()
        close(v, g["catkl_" + tag]); close(lq.grad, g["dlq_" + tag])
    lq = T(g["log_qy"])
    close(ck(lq, torch.log(torch.ones(1) / 5), batch_size=4), g["catkl_uniform"])
    close(ck(lq.view(1, 12, 5), lpy, batch_size=4), g["catkl_3d"])
    assert abs(float(ck(lq, lq, batch_size=4))) < 1e-7
    close(R.Entropy()(lq, batch_size=4), g["ent_bs"])
    close(R.Entropy()(lq, unit_average=True), g["ent_ua"])


def test_block_and_pyramids():
    g = load_golden("block")
    blk = load_params(R.Block(32, 16, 24), g)
    blk.train()
    x = T(g["x"]).requires_grad_()
    y = blk(x)
    (y * T(g["w"])).sum().backward()
    close(y, g["y"], rtol=1e-4, atol=1e-5)
    close(x.grad, g["dx"], rtol=1e-4, atol=1e-5)
    check_grads(blk, g, rtol=2e-4, atol=1e-5)
    for k, v in blk.state_dict().items():      # power-iteration buffers advanced identically
        close(v, g["post." + k], rtol=1e-4, atol=1e-6)
    close(blk(T(g["x"])), g["y2"], rtol=1e-4, atol=1e-5)
    blk.eval()
    close(blk(T(g["x"])), g["y_eval"], rtol=1e-4, atol=1e-5)

    for name, wrap in (("dlsrec", False), ("dlspri", True)):
        g = load_golden(name)
        m = load_params(R.DlsPyramid(64, 16, wrap=wrap), g)
        m.eval()
        out, acts = m(T(g["x"]))
        assert len(acts) == int(g["n_acts"])
        close(out, g["out"], rtol=1e-4, atol=1e-5)
        for i, a in enumerate(acts):
            close(a, g["act%d" % i], rtol=1e-4, atol=1e-5)


def test_dls_helpers():
    g = load_golden("dls_helpers")
    ts = [T(g[k]).requires_grad_() for k in ("m1", "m2", "l1", "l2")]
    kl = R.gaussian_analytical_kl(ts[0], ts[1], ts[2] * 0.3, ts[3] * 0.3)
    kl.backward()
    close(kl, g["kl"], rtol=1e-5)
    for n, t in zip(("dm1", "dm2", "dl1", "dl2"), ts):
        close(t.grad, g[n], rtol=1e-4, atol=1e-6)
    close(R.gaussian_analytical_kl_standard(T(g["m1"]), T(g["l1"]) * 0.3), g["kl_std"], rtol=1e-5)
    mm, mv = R.gaussian_merge(T(g["m1"]), T(g["m2"]), T(g["l1"]), T(g["l2"]))
    close(mm, g["merge_mean"], rtol=1e-4, atol=1e-6); close(mv, g["merge_logvar"], rtol=1e-4, atol=1e-6)
    v1, v2 = T(g["v1"]).requires_grad_(), T(g["v2"]).requires_grad_()
    cl = R.cons_loss(v1, v2, torch.arange(v1.size(0)))
    cl.backward()
    close(cl, g["cons"], rtol=1e-5)
    close(v1.grad, g["dv1"], rtol=1e-4, atol=1e-6); close(v2.grad, g["dv2"], rtol=1e-4, atol=1e-6)
    close(R.reparameterization(T(g["m1"]), T(g["l1"]), T(g["d"])), g["hash_rand"])
    close(R.reparameterization(T(g["m1"]), T(g["l1"]), T(g["u"])), g["dis_reparam"])
    steps = (0, 1, 100, 2500, 9000)
    close(np.array([R.kl_anneal_function("logistic", s, 0.0025, 2500) for s in steps]), g["anneal_logistic"])
    close(np.array([R.kl_anneal_function("linear", s, 0.0025, 2500) for s in steps]), g["anneal_linear"])


def test_den_auto_oracle_matches_reference():
    """oracle DeNAuto vs the golden made from the reference's own DeN_Auto (oracle/make_golden.py gen_den_auto): outputs,
    every gradient and the advanced power-iteration buffers, with the reference's random draws as inputs."""
    import types
    g = load_golden("den_auto")
    cfg = types.SimpleNamespace(expand_weight=0.5, converge_weight=0.1, ost=False, no_mutual_information=False)
    m = load_params(R.DeNAuto(16, 32, cfg), g)
    m.train()
    output = T(g["output"]).requires_grad_()
    act1 = [T(g["act1_%d" % i]).requires_grad_() for i in range(2)]
    act2 = [T(g["act2_%d" % i]).requires_grad_() for i in range(2)]
    z, rkl = m(output, act1, act2, False, [T(g["noise_%d" % i]) for i in range(2)], [T(g["gumbel_%d" % i]) for i in range(2)])
    (z.sum() * 0.01 + rkl).backward()
    close(z, g["z"], rtol=1e-5, atol=1e-6)
    close(rkl, g["rkl"], rtol=1e-5, atol=1e-6)
    close(output.grad, g["d_output"], rtol=1e-4, atol=1e-6)
    for i in range(2):
        close(act1[i].grad, g["d_act1_%d" % i], rtol=1e-4, atol=1e-6)
        close(act2[i].grad, g["d_act2_%d" % i], rtol=1e-4, atol=1e-6)
    n = 0
    for k, p in m.named_parameters():
        close(p.grad, g["g." + k], rtol=2e-4, atol=2e-6)
        n += 1
    assert n > 50
    for k, v in m.state_dict().items():
        if "post." + k in g:
            close(v, g["post." + k], rtol=1e-5, atol=1e-6)


def _dls_cfg(B, gir_weight=0.0):
    import types
    return types.SimpleNamespace(batch_size=B, expand_weight=0.5, converge_weight=0.1, cip_weight=1.0, gir_weight=gir_weight,
                                 ost=False)


def dls_auto_from_golden(cls, g, dev=None):
    """A DLSAuto (oracle or HIP) set up as the reference class was when the golden was made: gir_weight and n_batch
    (the density-estimate regulariser, dialog_models.py:3232-3236, :3311) come from the fixture."""
    m = load_params(cls(4, 16, _dls_cfg(10, float(g["gir_weight"]))), g)
    m.n_batch = torch.tensor(float(g["n_batch"]))
    return m if dev is None else m.to(dev)


def run_dls_auto_golden(module, g, T_, rtol, atol_rel):
    """Shared by the oracle test below and tests/test_gpu_dls.py: feed a DLSAuto with the inputs / draws of a dls_auto_*
    golden (made from the reference's own DLS_Auto class) and compare outputs and every gradient."""
    restarted, training = bool(g["restarted"]), bool(g["training"])
    module.train(training)
    ins = {k: T_(g[k]).requires_grad_() for k in ("output", "rec_output")}
    act1 = [T_(g["act1_%d" % i]).requires_grad_() for i in range(3)]
    act2 = [T_(g["act2_%d" % i]).requires_grad_() for i in range(3)]
    z, rkl = module(ins["output"], ins["rec_output"], act1, act2, restarted, noise=[T_(g["noise_%d" % i]) for i in range(3)])
    (z.sum() * 0.1 + rkl).backward()
    tol = lambda ref: dict(rtol=rtol, atol=atol_rel * float(np.abs(ref).max()) + 1e-7)
    close(z.cpu(), g["z"], **tol(g["z"]))
    close(rkl.cpu(), g["rkl"], rtol=rtol, atol=1e-6)
    n = 0
    for name, t in [("d_output", ins["output"]), ("d_rec_output", ins["rec_output"])] + \
            [("d_act1_%d" % i, act1[i]) for i in range(3)] + [("d_act2_%d" % i, act2[i]) for i in range(3)]:
        if name in g:
            close(t.grad.cpu(), g[name], **tol(g[name]))
            n += 1
    assert n >= 5
    for k, p in module.named_parameters():
        if "g." + k in g:
            close(p.grad.cpu(), g["g." + k], **tol(g["g." + k]))
            n += 1
    assert n > 40
    for k, v in module.state_dict().items():
        if "post." + k in g and training:
            close(v.cpu(), g["post." + k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["train", "restarted_train", "restarted_eval", "train_gir"])
def test_dls_auto_oracle_matches_reference_class(tag):
    """oracle DLSAuto vs the reference's own DLS_Auto.forward (dialog_models.py:3274-3352; golden from the class itself,
    constructor deviations written in oracle/make_golden.py gen_dls_auto): the composition is pinned, not only its parts."""
    g = load_golden("dls_auto_" + tag)
    run_dls_auto_golden(dls_auto_from_golden(R.DLSAuto, g), g, T, 2e-4, 2e-5)


def composed_step_inputs(g, T_):
    from pevae_amd.dls_vae import default_config
    V, E, H, L, B, Z, NU, NI = [int(x) for x in g["dims"]]
    stack = "dls" if "eps_0" in g else "gauss"
    cfg = default_config(vocab_size=V, embed_size=E, dec_cell_size=H, max_seq_len=L, n_user=NU, n_item=NI, latent_size=Z,
                         dropout=0.0, batch_size=B, latent_stack=stack, bos_id=4, eos_id=5, unk_id=1)
    feed = {"user_ids": T_(g["user_ids"]), "item_ids": T_(g["item_ids"]), "ratings": T_(g["ratings"]),
            "sentence_embeddings": T_(g["input_ids"]), "input_ids": T_(g["input_ids"]), "ans_ids": T_(g["ans_ids"])}
    eps = [T_(g["eps_%d" % i]) for i in range(3)] if stack == "dls" else T_(g["eps"])
    return cfg, feed, eps


@pytest.mark.parametrize("stack", ["gauss", "dls"])
def test_composed_step_oracle_matches_reference_modules(stack):
    """oracle/ref_step.py vs the same step composed from the reference's OWN classes (golden composed_step_*):
    losses and every parameter gradient -- a wiring error shared by the oracle step and pevae_amd/dls_vae.py would show here."""
    from oracle.ref_step import RefDLSVAE
    g = load_golden("composed_step_" + stack)
    cfg, feed, eps = composed_step_inputs(g, T)
    m = load_params(RefDLSVAE(cfg), g)
    m.train()
    loss = m(feed, eps=eps)
    m.valid_loss(loss).backward()
    close(loss.vae_nll, g["vae_nll"], rtol=1e-5)
    close(loss.vae_PPL, g["vae_ppl"], rtol=1e-5)
    close(loss.vae_rkl, g["vae_rkl"], rtol=1e-5, atol=1e-6)
    n = 0
    for k, p in m.named_parameters():
        if "g." + k in g:
            # hand-rolled recurrence cells vs ATen's fused GRU: 1e-4 of the tensor's largest entry
            # (+ 1e-6 absolute: cancellation residues in the deepest, 2-wide pyramid Block have |grad| ~ 2e-5)
            close(p.grad, g["g." + k], rtol=1e-3, atol=1e-4 * float(np.abs(g["g." + k]).max()) + 1e-6)
            n += 1
    assert n == len([k for k in g if k.startswith("g.")]) and n > 15
