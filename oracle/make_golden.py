#!/usr/bin/env python3
"""DEV-ONLY generator of tests/golden/*.npz.  Runs in the build container, never on the GPU box.

Imports the reference's own module classes from /root/reference (read-only) and records seeded
inputs, exported parameters, outputs and gradients for every hot-path module.
Two in-memory `sys.modules` stubs make the import possible without modifying any reference file:
  * `nltk` (models/utils.py:10-11 imports it; only tokenizer helpers use it, no arithmetic),
  * `numpy.lib.type_check` (stray unused import at models/enc2dec/decoders.py:2; gone in numpy 2).
`torch.Tensor.cuda` is patched to identity only while `DecoderRNN.__init__` runs, because of the
unconditional `torch.zeros(V).cuda()` at decoders.py:144.  No reference source is copied: the
output is data (inputs / parameters / expected outputs) only.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import types
import argparse

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    nltk = types.ModuleType("nltk")

    class _Tok:
        def __init__(self, *a, **k):
            pass

        def tokenize(self, x):
            return x.split() if isinstance(x, str) else " ".join(x)

    nltk.RegexpTokenizer = _Tok
    tok = types.ModuleType("nltk.tokenize")
    tb = types.ModuleType("nltk.tokenize.treebank")
    tb.TreebankWordDetokenizer = _Tok
    tok.treebank = tb
    nltk.tokenize = tok
    sys.modules.setdefault("nltk", nltk)
    sys.modules.setdefault("nltk.tokenize", tok)
    sys.modules.setdefault("nltk.tokenize.treebank", tb)
    tc = types.ModuleType("numpy.lib.type_check")
    tc._real_if_close_dispatcher = lambda *a, **k: None
    sys.modules.setdefault("numpy.lib.type_check", tc)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.enc2dec import encoders, decoders, base_modules   # noqa: E402
    from models import nn_lib, criterions, utils                   # noqa: E402
    return encoders, decoders, base_modules, nn_lib, criterions, utils


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def sd(module, prefix="p."):
    return {prefix + k: npy(v).copy() for k, v in module.state_dict().items()}


def grads(module, prefix="g."):
    return {prefix + k: npy(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    arrs = {k: npy(v) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrs)
    print("wrote", name, sum(v.size for v in arrs.values()), "elements")


class _CudaIdentity:
    def __enter__(self):
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._orig


def make_decoder(decoders, **kw):
    with _CudaIdentity():
        return decoders.DecoderRNN(**kw)


def gen_encoder(encoders):
    for tag, B, L, E, H, kw in (
            ("gru", 4, 7, 12, 16, {}),
            ("gru_mid", 6, 9, 20, 32, {}),
            ("gru_bi", 3, 5, 8, 16, dict(bidirection=True)),
            ("gru_2l", 3, 5, 8, 16, dict(n_layers=2)),
            ("lstm", 3, 5, 8, 16, dict(rnn_cell="lstm"))):
        torch.manual_seed(11)
        enc = encoders.EncoderRNN(E, H, **kw)
        x = torch.randn(B, L, E, requires_grad=True)
        dirs = 2 if kw.get("bidirection") else 1
        layers = kw.get("n_layers", 1)
        h0 = torch.randn(layers * dirs, B, H) * 0.5
        w_out = torch.randn(B, L, H * dirs)
        w_hid = torch.randn(layers * dirs, B, H)
        if kw.get("rnn_cell") == "lstm":
            c0 = torch.randn(layers * dirs, B, H) * 0.5
            out, (hn, cn) = enc(x, init_state=(h0, c0))
            loss = (out * w_out).sum() + (hn * w_hid).sum() + (cn * w_hid).sum() * 0.5
            extra = dict(c0=c0, cn=cn)
        else:
            out, hn = enc(x, init_state=h0)
            loss = (out * w_out).sum() + (hn * w_hid).sum()
            extra = {}
        loss.backward()
        out0, hn0 = enc(x.detach())          # no init state
        if kw.get("rnn_cell") == "lstm":
            hn0 = hn0[0]
        save("encoder_" + tag, x=x, h0=h0, w_out=w_out, w_hid=w_hid, out=out, hn=hn, loss=loss,
             dx=x.grad, out_nostate=out0, hn_nostate=hn0, **extra, **sd(enc), **grads(enc))
    # variable-length path (encoders.py:55-65)
    torch.manual_seed(12)
    enc = encoders.EncoderRNN(8, 16, variable_lengths=True)
    x = torch.randn(4, 6, 8)
    lens = [6, 5, 3, 2]
    out, hn = enc(x, input_lengths=lens)
    save("encoder_gru_varlen", x=x, lens=np.array(lens), out=out, hn=hn, **sd(enc))


def gen_utt_encoder(encoders):
    """RnnUttEncoder.forward (encoders.py:69-135) from the reference class: [B, n_ctx, n_tok] token ids -> one vector per
    utterance.  'attn': bidirectional GRU, extra per-utterance feature column, additive self-attention pooling (:117-122);
    'last': unidirectional GRU with a repeated initial state (:104-105), last state as the utterance vector (:123-126)."""
    for tag, kw, bidir in (("attn", dict(bidirection=True, use_attn=True, feat_size=1), True),
                           ("last", dict(bidirection=False, use_attn=False, feat_size=0), False)):
        torch.manual_seed(13)
        V, E, H, B, C, Lt = 23, 8, 16, 3, 2, 5
        enc = encoders.RnnUttEncoder(H, 0.0, rnn_cell="gru", vocab_size=V, embed_dim=E, **kw)
        utts = torch.randint(0, V, (B, C, Lt))
        feats = torch.randn(B, C) if kw["feat_size"] else None
        init = None if bidir else torch.randn(1, B, H) * 0.5
        utt, enc_outs, enc_last, attn = enc(utts, feats=feats, init_state=init, return_all=True)
        w = torch.randn_like(utt)
        loss = (utt * w).sum()
        loss.backward()
        extra = {}
        if feats is not None:
            extra["feats"] = feats
        if init is not None:
            extra["init_state"] = init
        if attn is not None:
            extra["attn"] = attn
        save("utt_encoder_" + tag, utterances=utts.numpy(), w=w, utt=utt, enc_outs=enc_outs, enc_last=enc_last, loss=loss,
             **extra, **sd(enc), **grads(enc))


def gen_encoder_bi_varlen(encoders):
    """Bidirectional GRU through the reference's pack_padded_sequence path (encoders.py:55-65): the reverse direction
    starts at each row's own last token."""
    torch.manual_seed(13)
    enc = encoders.EncoderRNN(8, 16, variable_lengths=True, bidirection=True)
    x = torch.randn(5, 7, 8, requires_grad=True)
    lens = [7, 6, 4, 2, 1]
    out, hn = enc(x, input_lengths=lens)
    w = torch.randn(out.shape)
    ((out * w).sum() + hn.sum()).backward()
    save("encoder_gru_bi_varlen", x=x, lens=np.array(lens), out=out, hn=hn, w=w, dx=x.grad, **sd(enc), **grads(enc))


def gen_encoder_lstm_varlen(encoders):
    """LSTM through the reference's pack_padded_sequence path, unidirectional and bidirectional."""
    for tag, bi in (("lstm_varlen", False), ("lstm_bi_varlen", True)):
        torch.manual_seed(14)
        enc = encoders.EncoderRNN(8, 16, rnn_cell="lstm", variable_lengths=True, bidirection=bi)
        x = torch.randn(5, 7, 8, requires_grad=True)
        lens = [7, 6, 4, 2, 1]
        out, (hn, cn) = enc(x, input_lengths=lens)
        w = torch.randn(out.shape)
        ((out * w).sum() + hn.sum() + 0.5 * cn.sum()).backward()
        save("encoder_" + tag, x=x, lens=np.array(lens), out=out, hn=hn, cn=cn, w=w, dx=x.grad, **sd(enc), **grads(enc))


def gen_decoder(decoders):
    V, L, E, H, B = 37, 7, 12, 16, 4
    for tag, kw, lat in (("gru", dict(rnn_cell="gru"), 0),
                         ("gru_latent", dict(rnn_cell="gru"), 5),
                         ("gru_temp", dict(rnn_cell="gru", softmax_temperature=0.7), 0),
                         ("lstm", dict(rnn_cell="lstm"), 0)):
        torch.manual_seed(21)
        emb = torch.nn.Embedding(V, E, padding_idx=0)
        dec = make_decoder(decoders, vocab_size=V, max_len=L, input_size=E + lat, hidden_size=H,
                           sos_id=1, eos_id=2, unk_id=3, use_gpu=False, embedding=emb, **kw)
        ids = torch.randint(4, V, (B, L))
        ids[:, 0] = 1
        ids[1, 5:] = 0
        ids[2, 3:] = 0
        h0 = torch.randn(1, B, H) * 0.5
        latent = torch.randn(B, lat) if lat else None
        w = torch.randn(B, L, V)
        if kw["rnn_cell"] == "lstm":
            c0 = torch.randn(1, B, H) * 0.5
            init = (h0.requires_grad_(), c0.requires_grad_())
        else:
            c0 = None
            init = h0.requires_grad_()
        logp, hid, ret = dec(B, ids, init, mode=decoders.TEACH_FORCE, latent_variable=latent)
        loss = (logp * w).sum()
        loss.backward()
        extra = {}
        if c0 is not None:
            extra.update(c0=c0, dc0=c0.grad, hid_c=hid[1])
            hid = hid[0]
        if latent is not None:
            extra["latent"] = latent
        save("decoder_tf_" + tag, ids=ids, h0=h0, w=w, logp=logp, hid=hid, loss=loss, dh0=h0.grad,
             length=np.array(ret["length"]), n_seq=np.array(len(ret["sequence"])),
             **extra, **sd(dec), **grads(dec))
    # free-running greedy (decoders.py:294-305), incl. mask_unk
    for tag, mask_unk in (("greedy", False), ("greedy_maskunk", True)):
        torch.manual_seed(22)
        emb = torch.nn.Embedding(V, E, padding_idx=0)
        dec = make_decoder(decoders, vocab_size=V, max_len=L, input_size=E, hidden_size=H, sos_id=1,
                           eos_id=2, unk_id=3, rnn_cell="gru", use_gpu=False, embedding=emb,
                           mask_unk=mask_unk)
        dec.eval()
        if mask_unk:
            with torch.no_grad():
                dec.project.bias[3] += 3.0      # make <unk> attractive so the mask matters
                dec.project.bias[2] += 0.25     # and EOS reachable
        ids = torch.randint(4, V, (B, L))
        h0 = torch.randn(1, B, H) * 0.5
        with torch.no_grad():
            logp, hid, ret = dec(B, ids, h0, mode=decoders.GEN, gen_type="greedy")
        seq = torch.cat(ret["sequence"], dim=1)
        save("decoder_gen_" + tag, ids=ids, h0=h0, logp=logp, hid=hid, seq=seq,
             length=np.array(ret["length"]), **sd(dec))


def gen_attention(decoders):
    B, Lo, Li, D, A = 3, 5, 6, 16, 16
    for mode in ("dot", "general", "cat"):
        torch.manual_seed(31)
        att = decoders.Attention(D, A, mode, project=True)
        out = torch.randn(B, Lo, D, requires_grad=True)
        ctx = torch.randn(B, Li, A, requires_grad=True)
        w = torch.randn(B, Lo, D)
        y, a = att(out, ctx)
        loss = (y * w).sum()
        loss.backward()
        save("attention_" + mode, out=out, ctx=ctx, w=w, y=y, attn=a, loss=loss, dout=out.grad,
             dctx=ctx.grad, **sd(att), **grads(att))


def gen_connectors(nn_lib):
    torch.manual_seed(41)
    B, D, Z = 6, 16, 8
    h2g = nn_lib.Hidden2Gaussian(D, Z)
    x = torch.randn(1, B, D, requires_grad=True)
    wm, wl = torch.randn(B, Z), torch.randn(B, Z)
    mu, lv = h2g(x)
    ((mu * wm).sum() + (lv * wl).sum()).backward()
    save("hidden2gaussian", x=x, wm=wm, wl=wl, mu=mu, logvar=lv, dx=x.grad, **sd(h2g), **grads(h2g))

    h2g = nn_lib.Hidden2Gaussian(D, Z, is_lstm=True)
    h, c = torch.randn(1, B, D), torch.randn(1, B, D)
    mu, lv = h2g((h, c))
    save("hidden2gaussian_lstm", h=h, c=c, mu=mu, logvar=lv, **sd(h2g))

    lc = nn_lib.LinearConnector(Z, D, False)
    z = torch.randn(B, Z, requires_grad=True)
    w = torch.randn(1, B, D)
    y = lc(z)
    (y * w).sum().backward()
    save("linear_connector", z=z, w=w, y=y, dz=z.grad, **sd(lc), **grads(lc))

    lc = nn_lib.LinearConnector(Z, D, True)
    yh, yc = lc(z.detach())
    save("linear_connector_lstm", z=z, yh=yh, yc=yc, **sd(lc))

    gc = nn_lib.GumbelConnector()
    logits = torch.randn(12, 5, requires_grad=True)
    for tag, kw in (("soft", {}), ("hard", dict(hard=True, return_max_id=True)),
                    ("st", dict(straight_through=True))):
        torch.manual_seed(42)
        u = torch.rand(logits.size())           # the draw the reference will make (nn_lib.py:231)
        torch.manual_seed(42)
        res = gc(logits, 0.7, False, **kw)
        y, ids = res if isinstance(res, tuple) else (res, None)
        w = torch.randn(12, 5)
        logits.grad = None
        (y * w).sum().backward()
        extra = dict(ids=ids) if ids is not None else {}
        save("gumbel_" + tag, logits=logits, u=u, w=w, y=y, dlogits=logits.grad, tau=np.array(0.7),
             **extra)

    gr = nn_lib.GreedyConnector()
    y, ids = gr(logits.detach(), False, return_max_id=True)
    save("greedy_connector", logits=logits, y=y, ids=ids)


def gen_criterions(criterions):
    torch.manual_seed(51)
    B, L, V = 4, 7, 37
    logits = torch.randn(B, L, V)
    logp = torch.log_softmax(logits, -1).requires_grad_()
    labels = torch.randint(1, V, (B, L))
    labels[0, 5:] = 0
    labels[3, 2:] = 0
    res = {}
    for avg in ("seq", "word", None):
        crit = criterions.NLLEntropy(0, None, avg_type=avg)
        logp.grad = None
        v = crit(logp, labels)
        v.backward()
        res["nll_%s" % avg] = v
        res["dlogp_%s" % avg] = logp.grad.clone()
    ppl = criterions.Perplexity(0, None)
    logp.grad = None
    v = ppl(logp, labels)
    v.backward()
    res["ppl"] = v
    res["dlogp_ppl"] = logp.grad.clone()
    save("criterions_nll", logp=logp, labels=labels, **res)

    Z = 8
    mq, lq, mp, lp = [torch.randn(6, Z, requires_grad=True) for _ in range(4)]
    res = {}
    for ua in (False, True):
        for t in (mq, lq, mp, lp):
            t.grad = None
        v = criterions.NormKLLoss(unit_average=ua)(mq, lq * 0.5, mp, lp * 0.5)
        v.backward()
        res["kl_ua%d" % ua] = v
        for n, t in (("mq", mq), ("lq", lq), ("mp", mp), ("lp", lp)):
            res["d%s_ua%d" % (n, ua)] = t.grad.clone()
    res["kl_same"] = criterions.NormKLLoss()(mq, lq, mq, lq)
    save("criterions_normkl", mq=mq, lq=lq, mp=mp, lp=lp, **res)

    lq = torch.log_softmax(torch.randn(12, 5), -1).requires_grad_()
    lpy = torch.log_softmax(torch.randn(12, 5), -1)
    ck = criterions.CatKLLoss()
    res = {}
    for tag, kw in (("bs", dict(batch_size=4)), ("ua", dict(unit_average=True)),
                    ("rows", dict(average=False))):
        lq.grad = None
        v = ck(lq, lpy, **kw)
        v.sum().backward()
        res["catkl_" + tag] = v
        res["dlq_" + tag] = lq.grad.clone()
    res["catkl_uniform"] = ck(lq, torch.log(torch.ones(1) / 5), batch_size=4)
    res["catkl_3d"] = ck(lq.view(1, 12, 5), lpy, batch_size=4)
    ent = criterions.Entropy()
    res["ent_bs"] = ent(lq, batch_size=4)
    res["ent_ua"] = ent(lq, unit_average=True)
    save("criterions_cat", log_qy=lq, log_py=lpy, **res)


def gen_dialog_pieces(utils):
    """Block / DlsRec / DlsPri / static helpers of DLS_Auto.  dialog_models imports transformers etc;
    it imports on CPU with the stubs already installed."""
    from models import dialog_models as dm
    torch.manual_seed(61)
    B = 6
    blk = dm.Block(32, 16, 24)
    blk.train()
    x = torch.randn(B, 32, requires_grad=True)
    w = torch.randn(B, 16)
    pre = {k: npy(v).copy() for k, v in blk.state_dict().items()}
    y = blk(x)
    (y * w).sum().backward()
    post = sd(blk, "post.")
    y2 = blk(x.detach())        # second training forward: power iteration advanced once more
    blk.eval()
    y_eval = blk(x.detach())
    save("block", x=x, w=w, y=y, y2=y2, y_eval=y_eval, dx=x.grad,
         **{"p." + k: v for k, v in pre.items()}, **post, **grads(blk))

    for name, cls in (("dlsrec", dm.DlsRec), ("dlspri", dm.DlsPri)):
        torch.manual_seed(62)
        m = cls(64, 16, None)
        m.eval()
        x = torch.randn(B, 1, 64)
        out, acts = m(x)
        save(name, x=x, out=out, n_acts=np.array(len(acts)),
             **{"act%d" % i: a for i, a in enumerate(acts)}, **sd(m))

    torch.manual_seed(63)
    m1, m2, l1, l2 = [torch.randn(B, 8, requires_grad=True) for _ in range(4)]
    kl = dm.DLS_Auto.gaussian_analytical_kl(m1, m2, l1 * 0.3, l2 * 0.3)
    kl.backward()
    kls = dm.DLS_Auto.gaussian_analytical_kl_standard(m1.detach(), l1.detach() * 0.3)
    mm, mv = dm.DLS_Auto.gaussian_merge(m1.detach(), m2.detach(), l1.detach(), l2.detach())
    v1 = torch.randn(B, 12, requires_grad=True)
    v2 = torch.randn(B, 12, requires_grad=True)
    cl = dm.DLS_Auto.cons_loss(v1, v2, torch.arange(B))
    cl.backward()
    d = torch.randn(B, 8)
    hr = dm.hash_rand(m1.detach(), l1.detach(), d)
    torch.manual_seed(64)
    dr = dm.DLS_Auto.dis_reparameterization(m1.detach(), l1.detach())
    torch.manual_seed(64)
    u = torch.rand_like(l1)
    anneal = np.array([utils.kl_anneal_function("logistic", s, 0.0025, 2500) for s in (0, 1, 100, 2500, 9000)])
    anneal_lin = np.array([utils.kl_anneal_function("linear", s, 0.0025, 2500) for s in (0, 1, 100, 2500, 9000)])
    save("dls_helpers", m1=m1, m2=m2, l1=l1, l2=l2, kl=kl, dm1=m1.grad, dm2=m2.grad, dl1=l1.grad,
         dl2=l2.grad, kl_std=kls, merge_mean=mm, merge_logvar=mv, v1=v1, v2=v2, cons=cl, dv1=v1.grad,
         dv2=v2.grad, d=d, hash_rand=hr, u=u, dis_reparam=dr, anneal_logistic=anneal,
         anneal_linear=anneal_lin)


def gen_den_auto():
    """DeN_Auto (dialog_models.py:2611-2764), training branch, from the reference class itself (it constructs on CPU).
    Its random draws (torch.rand_like at :2658, exponential_ at :2682) come from the seeded CPU generator; the same
    stream is re-drawn here in the same order and stored, so the HIP path and the oracle take the draws as inputs."""
    from models import dialog_models as dm
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_modules as R
    cfg = types.SimpleNamespace(expand_weight=0.5, converge_weight=0.1, ost=False, no_mutual_information=False)
    torch.manual_seed(71)
    B, d0, dmax, nlev = 8, 16, 32, 2
    m = dm.DeN_Auto(d0, dmax, cfg)
    m.train()
    pre = {k: npy(v).copy() for k, v in m.state_dict().items()}
    widths = [2 * d0 * 2 ** i for i in range(nlev)]              # act[-1], act[-2]
    output = torch.randn(B, d0, requires_grad=True)
    act1 = [torch.randn(B, w, requires_grad=True) for w in reversed(widths)]
    act2 = [torch.randn(B, w, requires_grad=True) for w in reversed(widths)]
    torch.manual_seed(72)
    z, rkl = m(output, act1, act2, None, False)
    (z.sum() * 0.01 + rkl).backward()
    torch.manual_seed(72)
    noise, gumbels = [], []
    for w in widths:
        noise.append(torch.rand(B, w))
        gumbels.append(-torch.empty(B, B).exponential_().log())
    # the re-drawn stream reproduces the reference: the oracle restatement fed with it must agree
    o = R.DeNAuto(d0, dmax, cfg)
    o.load_state_dict({k: torch.from_numpy(v) for k, v in pre.items()})
    o.train()
    z2, rkl2 = o(output.detach(), [a.detach() for a in act1], [a.detach() for a in act2], False, noise, gumbels)
    assert torch.allclose(z2, z, rtol=1e-5, atol=1e-6) and abs(float(rkl2) - float(rkl)) < 1e-5 * abs(float(rkl)), \
        (float((z2 - z).abs().max()), float(rkl2), float(rkl))
    arrs = dict(output=output, z=z, rkl=rkl, d_output=output.grad)
    for i in range(nlev):
        arrs.update({"act1_%d" % i: act1[i], "act2_%d" % i: act2[i], "d_act1_%d" % i: act1[i].grad,
                     "d_act2_%d" % i: act2[i].grad, "noise_%d" % i: noise[i], "gumbel_%d" % i: gumbels[i]})
    post = {k: v for k, v in sd(m, "post.").items() if k.endswith("weight_u") or k.endswith("weight_v")}
    save("den_auto", **arrs, **{"p." + k: v for k, v in pre.items()}, **post, **grads(m))


class _CpuCurrentDevice:
    """DLS_Auto.__init__ places its (unused here) Normal priors with `.to(torch.cuda.current_device())`
    (dialog_models.py:3198,3203); returning 'cpu' for the duration of the constructor lets the class be built without a GPU."""

    def __enter__(self):
        self._orig = torch.cuda.current_device
        torch.cuda.current_device = lambda: "cpu"

    def __exit__(self, *exc):
        torch.cuda.current_device = self._orig


def _dls_cfg(B, gir_weight=0.0):
    # cip_weight / gir_weight are read by DLS_Auto.forward (:3287, :3311) but defined by no parser.
    # gir_weight = 0 switches the density-estimate regulariser off in value and gradient (it raises in the reference as
    # written: torch.log of the Python int n_batch = -1, :3169, :3235); the 'train_gir' fixture switches it on.
    return types.SimpleNamespace(batch_size=B, expand_weight=0.5, converge_weight=0.1, cip_weight=1.0, gir_weight=gir_weight,
                                 ost=False)


def _redraw_uniform(seed, shapes):
    torch.manual_seed(seed)
    return [torch.rand(*s) for s in shapes]


def gen_dls_auto():
    """DLS_Auto.forward (dialog_models.py:3274-3352) from the reference class itself, all three branches.  Deviations
    needed to run it, both outside the arithmetic the HIP path reproduces: the constructor runs with
    torch.cuda.current_device patched to 'cpu' (_CpuCurrentDevice), and n_batch (:3169, the int -1) is set to
    tensor(B) so that density_estimate's torch.log (:3235) has a tensor argument; with gir_weight = 0 that term then
    contributes exactly 0.  The uniform draws of dis_reparameterization (:3207) come from the seeded CPU generator and are
    re-drawn in the same order and stored."""
    from models import dialog_models as dm
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ref_modules as R
    B, d0, dmax = 10, 4, 16
    for tag, restarted, training in (("train", False, True), ("restarted_train", True, True), ("restarted_eval", True, False),
                                     ("train_gir", False, True)):
        cfg = _dls_cfg(B, gir_weight=0.25 if tag == "train_gir" else 0.0)      # train_gir: density-estimate term on
        torch.manual_seed(81)
        with _CpuCurrentDevice():
            m = dm.DLS_Auto(d0, dmax, cfg)
        m.n_batch = torch.tensor(float(B))
        m.train(training)
        pre = {k: npy(v).copy() for k, v in m.state_dict().items() if k != "batch_arrange"}
        widths = [2 * d0 * 2 ** i for i in range(3)]                       # act[-1], act[-2], act[-3]
        output = torch.randn(B, d0, requires_grad=True)
        rec_output = torch.randn(B, d0, requires_grad=True)
        act1 = [torch.randn(B, w, requires_grad=True) for w in reversed(widths)]
        act2 = [torch.randn(B, w, requires_grad=True) for w in reversed(widths)]
        torch.manual_seed(82)
        z, rkl = m(output, rec_output, act1, act2, restarted)
        (z.sum() * 0.1 + rkl).backward()
        noise = _redraw_uniform(82, [(B, w) for w in widths])
        o = R.DLSAuto(d0, dmax, cfg)
        o.load_state_dict({k: torch.from_numpy(v) for k, v in pre.items()})
        o.n_batch = torch.tensor(float(B))
        o.train(training)
        z2, rkl2 = o(output.detach(), rec_output.detach(), [a.detach() for a in act1], [a.detach() for a in act2],
                     restarted, noise=noise)
        assert torch.allclose(z2, z, rtol=1e-4, atol=2e-5) and abs(float(rkl2) - float(rkl)) <= 1e-5 * abs(float(rkl)), \
            (tag, float((z2 - z).abs().max()), float(rkl2), float(rkl))
        arrs = dict(output=output, rec_output=rec_output, z=z, rkl=rkl, d_output=output.grad, d_rec_output=rec_output.grad,
                    restarted=np.array(restarted), training=np.array(training), gir_weight=np.array(cfg.gir_weight),
                    n_batch=np.array(float(B)))
        for i in range(3):
            arrs.update({"act1_%d" % i: act1[i], "act2_%d" % i: act2[i], "noise_%d" % i: noise[i]})
            for nm, a in (("d_act1_%d" % i, act1[i]), ("d_act2_%d" % i, act2[i])):
                if a.grad is not None:
                    arrs[nm] = a.grad
        post = {k: v for k, v in sd(m, "post.").items() if k.endswith("weight_u") or k.endswith("weight_v")}
        save("dls_auto_" + tag, **arrs, **{"p." + k: v for k, v in pre.items()}, **post,
             **{k: v for k, v in grads(m).items() if "batch_arrange" not in k})


def gen_composed_step(encoders, decoders, nn_lib, criterions):
    """The composed DLS_VAE training step of pevae_amd/dls_vae.py built from the REFERENCE's own classes (SURVEY 3.2 /
    8c): nn.Embedding tables (dialog_models.py:468-502), EncoderRNN (:530), DecoderRNN sharing the word embedding
    (:560-570), NLLEntropy / Perplexity (:634-637) and, per latent stack,
      'gauss': Hidden2Gaussian heads, z = mu + exp(logvar) * eps (:191-203), NormKLLoss, LinearConnector (:626);
      'dls'  : DlsRec / DlsPri / DLS_Auto / x_init_connector = Block -> Block (:3401-3417, :3529-3538).
    Data flow of FS_VAE.forward (:755-818), loss = vae_nll + vae_rkl (:3451-3465).  What is NOT the reference's: the
    composition itself (its DLS_VAE cannot be constructed), prior_fc + tanh as the condition encoder.
    Parameters are stored under the state_dict keys of pevae_amd.dls_vae.DLSVAE."""
    from models import dialog_models as dm
    V, E, H, L, B, Z, NU, NI = 97, 12, 32, 7, 6, 8, 11, 9
    for stack in ("gauss", "dls"):
        torch.manual_seed(91)
        cfg = _dls_cfg(B)
        x_embedding = torch.nn.Embedding(V, E, padding_idx=0)
        user_embedding, item_embedding = torch.nn.Embedding(NU, H // 2), torch.nn.Embedding(NI, H // 2)
        rating_embedding = torch.nn.Embedding(5, H // 2)
        x_encoder = encoders.EncoderRNN(E, H, dropout_p=0.0, rnn_cell="gru", variable_lengths=False)
        x_decoder = make_decoder(decoders, vocab_size=V, max_len=L, input_size=E, hidden_size=H, sos_id=4, eos_id=5,
                                 unk_id=1, n_layers=1, rnn_cell="gru", input_dropout_p=0.0, dropout_p=0.0,
                                 use_attention=False, use_gpu=False, embedding=x_embedding)
        prior_fc = torch.nn.Linear(3 * (H // 2), H)
        mods = dict(x_embedding=x_embedding, user_embedding=user_embedding, item_embedding=item_embedding,
                    rating_embedding=rating_embedding, x_encoder=x_encoder, x_decoder=x_decoder, prior_fc=prior_fc)
        if stack == "gauss":
            mods.update(rec_head=nn_lib.Hidden2Gaussian(H, Z), pri_head=nn_lib.Hidden2Gaussian(H, Z),
                        x_init_connector=nn_lib.LinearConnector(Z, H, False))
        else:
            with _CpuCurrentDevice():
                sup = dm.DLS_Auto(H // 32, H // 8, cfg)
            sup.n_batch = torch.tensor(float(B))
            mods.update(rec_encoder=dm.DlsRec(H, H // 16, cfg), pri_encoder=dm.DlsPri(H, H // 16, cfg), sup_decoder=sup,
                        x_init_connector=torch.nn.Sequential(dm.Block(H // 4, H // 2, H // 2), dm.Block(H // 2, H, H)))
        model = torch.nn.ModuleDict(mods)
        model.train()
        nll_loss, ppl = criterions.NLLEntropy(0, None), criterions.Perplexity(0, None)
        kl_loss = criterions.NormKLLoss(unit_average=False)
        pre = {k: npy(v).copy() for k, v in model.state_dict().items() if "batch_arrange" not in k}
        # batch in the format of YELPDataLoader.batcher (data_loaders.py:704-715)
        lens = torch.randint(3, L + 1, (B,))
        ids = torch.randint(6, V, (B, L))
        ids = torch.where(torch.arange(L).unsqueeze(0) < lens.unsqueeze(1), ids, torch.zeros_like(ids))
        ids[:, 0] = 4
        ids[torch.arange(B), lens - 1] = 5
        ans = torch.cat([ids[:, 1:], torch.zeros(B, 1, dtype=torch.long)], 1)
        user, item = torch.randint(0, NU, (B,)), torch.randint(0, NI, (B,))
        ratings = torch.randint(0, 5, (B,)).float()
        # -- the step ------------------------------------------------------------------------------------------
        u, it = user_embedding(user), item_embedding(item)                                  # :758-759
        r = rating_embedding(ratings.ge(3).int())                                           # :760
        enc_outs, enc_last = x_encoder(x_embedding(ids))                                    # :765
        h = enc_last.transpose(0, 1).contiguous().squeeze(1)                                # :766
        prior_h = torch.tanh(prior_fc(torch.cat([u, it, r], -1)))
        if stack == "gauss":
            mu_q, lv_q = mods["rec_head"](h)
            mu_p, lv_p = mods["pri_head"](prior_h)
            eps = torch.randn(B, Z)
            z = eps * torch.exp(lv_q) + mu_q                                                # :191-203
            rkl = kl_loss(mu_q, lv_q, mu_p, lv_p)
            dec_init = mods["x_init_connector"](z)                                          # :711
            noise = {"eps": eps}
        else:
            z_pri, pri_state = mods["pri_encoder"](prior_h.unsqueeze(1))                    # :3529 (output[:, 0])
            z_rec, rec_state = mods["rec_encoder"](h.unsqueeze(1))                          # :3530
            torch.manual_seed(92)
            z, rkl = mods["sup_decoder"](z_pri, z_rec, rec_state, pri_state, False)         # :3535
            draws = _redraw_uniform(92, [(B, H // 16), (B, H // 8), (B, H // 4)])
            noise = {"eps_%d" % i: d for i, d in enumerate(draws)}
            dec_init = mods["x_init_connector"](z).unsqueeze(0)                             # :3473
        logp, _, _ = x_decoder(B, ids, dec_init, mode=decoders.TEACH_FORCE)                 # :712
        vae_nll, vae_ppl = nll_loss(logp, ans), ppl(logp, ans)                              # :790-791
        (vae_nll + rkl).backward()                                                          # :3451-3465
        g = {"g." + k: npy(p.grad) for k, p in model.named_parameters() if p.grad is not None and "batch_arrange" not in k}
        post = {"post." + k: npy(v).copy() for k, v in model.state_dict().items()
                if k.endswith("weight_u") or k.endswith("weight_v")}
        save("composed_step_" + stack, input_ids=ids, ans_ids=ans, user_ids=user, item_ids=item, ratings=ratings,
             vae_nll=vae_nll, vae_ppl=vae_ppl, vae_rkl=rkl, dims=np.array([V, E, H, L, B, Z, NU, NI]),
             **noise, **{"p." + k: v for k, v in pre.items()}, **post, **g)


def gen_yelp_data_path():
    """Real data path of the step from the reference's own code: yelp._process_reivews (corpora.py:609-622),
    yelp._build_vocab (:624-651) and YELPDataLoader.batcher (data_loaders.py:636-715).  The `yelp` class cannot be
    constructed (it unpickles corpus files that do not ship), so its two methods are called on a stand-in object that
    carries the attributes they read; the tokenizer is nltk.RegexpTokenizer(r'\\w+|<sil>|[^\\w\\s]+').tokenize in the
    reference (utils.py:241-242) -- nltk is absent here, and with gaps=False that class is re.findall of the same pattern,
    which is what the stand-in uses.  Output: tests/golden/yelp_data_path.json (raw reviews, processed rows, vocabulary,
    collated batches as lists)."""
    import json
    import random
    import re
    from models.dataset import corpora, data_loaders
    rnd = random.Random(5)
    pool = ["good", "food", "the", "service", "was", "great", "slow", "place", "pizza", "love", "it's", "not", "bad", "!", ",",
            "very", "tasty", "staff", "friendly", "never", "again", "<sil>", "5-star", "wait...", "ok"]
    def review():
        return " ".join(rnd.choice(pool) for _ in range(rnd.randint(1, 14)))
    raw = {name: [(rnd.randrange(40), rnd.randrange(25), rnd.randrange(7), review(), "unused", rnd.randint(1, 5))
                  for _ in range(n)] for name, n in (("train", 40), ("valid", 12), ("test", 9), ("pretrain", 7))}
    obj = types.SimpleNamespace(tokenize=re.compile(r"\w+|<sil>|[^\w\s]+").findall)
    torch.manual_seed(101)
    rows = {name: corpora.yelp._process_reivews(obj, data) for name, data in raw.items()}
    obj.train_corpus, obj.valid_corpus, obj.test_corpus, obj.pretrain_corpus = (rows[k] for k in ("train", "valid", "test", "pretrain"))
    corpora.yelp._build_vocab(obj, 30)                      # cut-off below the raw vocabulary size: OOV words exist
    batches = []
    for max_seq_len, sel in ((8, range(0, 10)), (8, range(10, 13)), (16, range(20, 27)), (5, range(30, 40))):
        b = data_loaders.YELPDataLoader.batcher(obj.rev_vocab, max_seq_len, "cpu")([rows["train"][i] for i in sel])
        batches.append(dict(max_seq_len=max_seq_len, rows=list(sel), out={k: v.tolist() for k, v in b.items()}))
    out = dict(raw={k: [list(r) for r in v] for k, v in raw.items()},
               rows={k: [list(r) for r in v] for k, v in rows.items()},
               max_vocab_cnt=30, vocab=obj.vocab, unk_id=obj.unk_id, batches=batches)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "yelp_data_path.json"), "w") as fh:
        json.dump(out, fh)
    print("wrote yelp_data_path.json:", len(obj.vocab), "vocabulary entries,", len(batches), "batches")


def gen_decoder_sample_beam(decoders):
    """Free-running `sample` and `beam` decoding (decoders.py:255-270, :294-328).
    sample: the reference run as is (seeded).  torch.multinomial(probs, 1) on the CPU draws q ~ Exp(1) per entry and returns
    argmax(probs / q); the same stream is re-drawn here (one [B, V] exponential_ per token, in order), checked to give
    the reference's symbols, and stored -- the HIP path and the oracle take it as input.
    beam: the reference's own forward() RAISES on this torch at the back-tracking step (torch.cat of 0-d tensors,
    :318-322; and top_id.div at :267 is a true division since torch 1.5).  The golden therefore drives the reference's own
    forward_step (:171-194) per token and restates decode()'s beam branch (:257-270) and the back-tracking (:307-328) with
    floor division -- the deviation DESIGN.md section 1 lists.  As in the reference, the decoder state is NOT re-ordered by
    the back pointers and lengths are taken over all B * beam raw symbol rows."""
    V, L, E, H, B, beam = 37, 7, 12, 16, 4, 3
    torch.manual_seed(23)
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = make_decoder(decoders, vocab_size=V, max_len=L, input_size=E, hidden_size=H, sos_id=1, eos_id=2, unk_id=3,
                       rnn_cell="gru", use_gpu=False, embedding=emb)
    dec.eval()
    with torch.no_grad():
        dec.project.bias[2] += 1.0                  # EOS reachable
    ids = torch.randint(4, V, (B, L))
    h0 = torch.randn(1, B, H) * 0.5
    # ---- sample ----------------------------------------------------------------------------------------------------
    torch.manual_seed(24)
    with torch.no_grad():
        logp, hid, ret = dec(B, ids, h0, mode=decoders.GEN, gen_type="sample")
    seq = torch.cat(ret["sequence"], dim=1)
    torch.manual_seed(24)
    qs = torch.stack([torch.empty(B, V).exponential_() for _ in range(L)])
    again = torch.stack([(torch.exp(logp[:, t]) / qs[t]).argmax(-1) for t in range(L)], 1)
    assert torch.equal(again, seq), "torch.multinomial's CPU stream is not exp-trick argmax on this torch build"
    save("decoder_gen_sample", ids=ids, h0=h0, logp=logp, hid=hid, seq=seq, length=np.array(ret["length"]), q=qs, **sd(dec))
    # ---- beam ------------------------------------------------------------------------------------------------------
    with torch.no_grad():
        inp = torch.full((B * beam, 1), 1, dtype=torch.long)                   # :216-219 sos_id
        hidden = dec.repeat_state(h0, B, beam)                                 # :225-227
        outs, symbols_all, back_all, cum = [], [], [], None
        lengths = np.array([L] * B * beam)
        for di in range(L):
            step_out, hidden, _ = dec.forward_step(inp, hidden, None)
            outs.append(step_out)
            sl = step_out.squeeze(1)
            score = sl.view(B, -1)[:, 0:V] if di == 0 else (cum + sl).view(B, -1)
            top_v, top_id = score.topk(beam)
            back_all.append(torch.div(top_id, V, rounding_mode="floor").view(-1, 1))
            sym = top_id.fmod(V).view(-1, 1)
            cum = top_v.view(-1, 1)
            symbols_all.append(sym)
            eos = sym.eq(2).view(-1).numpy()
            lengths[((lengths > di) & eos) != 0] = len(symbols_all)          # :276-280
            inp = sym
        best = cum.view(-1, beam).topk(1)[1].view(-1)
        final = []
        for sym, bp in zip(symbols_all[::-1], back_all[::-1]):
            final.append(sym.view(-1, beam)[torch.arange(B), best].unsqueeze(1))
            best = bp.view(-1, beam)[torch.arange(B), best]
        final = final[::-1]
    save("decoder_gen_beam", ids=ids, h0=h0, logp=torch.cat(outs, 1), hid=hidden, seq=torch.cat(final, 1),
         raw_symbols=torch.cat(symbols_all, 1), length=lengths, beam=np.array(beam), **sd(dec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(False)
    encoders, decoders, base_modules, nn_lib, criterions, utils = import_reference()
    todo = dict(encoder=lambda: gen_encoder(encoders), utt_encoder=lambda: gen_utt_encoder(encoders), encoder_bi_varlen=lambda: gen_encoder_bi_varlen(encoders),
                encoder_lstm_varlen=lambda: gen_encoder_lstm_varlen(encoders), decoder=lambda: gen_decoder(decoders),
                attention=lambda: gen_attention(decoders), connectors=lambda: gen_connectors(nn_lib),
                criterions=lambda: gen_criterions(criterions),
                dialog=lambda: gen_dialog_pieces(utils), den=gen_den_auto, dls_auto=gen_dls_auto, yelp=gen_yelp_data_path,
                sample_beam=lambda: gen_decoder_sample_beam(decoders),
                composed=lambda: gen_composed_step(encoders, decoders, nn_lib, criterions))
    for k, fn in todo.items():
        if not args.only or k in args.only.split(","):
            fn()


if __name__ == "__main__":
    main()
