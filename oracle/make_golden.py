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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(False)
    encoders, decoders, base_modules, nn_lib, criterions, utils = import_reference()
    todo = dict(encoder=lambda: gen_encoder(encoders), encoder_bi_varlen=lambda: gen_encoder_bi_varlen(encoders),
                encoder_lstm_varlen=lambda: gen_encoder_lstm_varlen(encoders), decoder=lambda: gen_decoder(decoders),
                attention=lambda: gen_attention(decoders), connectors=lambda: gen_connectors(nn_lib),
                criterions=lambda: gen_criterions(criterions),
                dialog=lambda: gen_dialog_pieces(utils), den=gen_den_auto)
    for k, fn in todo.items():
        if not args.only or k in args.only.split(","):
            fn()


if __name__ == "__main__":
    main()
