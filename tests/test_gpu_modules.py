"""GPU parity tests, module level: the drop-in classes (same constructors / forward signatures / state_dict
keys as the reference) are loaded with the parameters exported from the reference and compared with the golden
outputs and gradients the reference produced (tests/golden/*.npz, made by oracle/make_golden.py).

Tolerances: fp32 kernels (connectors, criterions, attention) 1e-4; modules whose MFMA operands are bf16
(GRU recurrence, vocabulary projection) 3e-2 absolute on O(1) activations / log-probs and 4e-2 of the largest
entry on gradients; symbols / indices / lengths bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, load_params


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.array(a))
    return t if dev is None else t.to(dev)


def close(a, b, rtol=1e-4, atol=1e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy(), np.asarray(b), rtol=rtol, atol=atol)


def relmax(a, b):
    a = a.detach().cpu().double().numpy()
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def check_grads(module, g, tol):
    n = 0
    for k, p in module.named_parameters():
        if "g." + k in g:
            assert relmax(p.grad, g["g." + k]) < tol, k
            n += 1
    assert n > 0


@pytest.mark.parametrize("tag,kw", [("gru", {}), ("gru_mid", {}), ("gru_bi", dict(bidirection=True)),
                                    ("gru_2l", dict(n_layers=2))])
def test_encoder_rnn(dev, tag, kw):
    from pevae_amd.enc2dec.encoders import EncoderRNN
    from pevae_amd import ops
    g = load_golden("encoder_" + tag)
    B, L, E = g["x"].shape
    H = g["h0"].shape[-1]
    enc = load_params(EncoderRNN(E, H, **kw), g).to(dev)
    x = T(g["x"], dev).requires_grad_()
    out, hn = enc(x, init_state=T(g["h0"], dev))
    ((out * T(g["w_out"], dev)).sum() + (hn * T(g["w_hid"], dev)).sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    close(out, g["out"], rtol=0, atol=3e-2)
    close(hn, g["hn"], rtol=0, atol=3e-2)
    assert relmax(x.grad, g["dx"]) < 4e-2
    check_grads(enc, g, 4e-2)
    out0, hn0 = enc(T(g["x"], dev))
    close(out0, g["out_nostate"], rtol=0, atol=3e-2)
    close(hn0, g["hn_nostate"], rtol=0, atol=3e-2)


@pytest.mark.parametrize("tag", ["attn", "last"])
def test_utt_encoder(dev, tag):
    """RnnUttEncoder (HIP) vs the golden from the reference class (encoders.py:69-135): bidirectional GRU + feature column +
    attention pooling, and unidirectional GRU with a repeated initial state + last-state pooling."""
    from pevae_amd.enc2dec.encoders import RnnUttEncoder
    from pevae_amd import ops
    from test_oracle_golden import _utt_encoder_case
    g = load_golden("utt_encoder_" + tag)
    enc, feats, init = _utt_encoder_case(RnnUttEncoder, g, tag, lambda a: T(a, dev))
    enc = enc.to(dev)
    utt, enc_outs, enc_last, attn = enc(T(g["utterances"], dev), feats=feats, init_state=init, return_all=True)
    (utt * T(g["w"], dev)).sum().backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    close(utt, g["utt"], rtol=0, atol=3e-2)
    close(enc_outs, g["enc_outs"], rtol=0, atol=3e-2)
    close(enc_last, g["enc_last"], rtol=0, atol=3e-2)
    if tag == "attn":
        close(attn, g["attn"], rtol=0, atol=2e-2)
    n = 0
    for k, p in enc.named_parameters():
        if k == "query.bias":            # a softmax is shift invariant: this gradient is zero, the golden holds its rounding
            assert float(p.grad.abs().max()) < 1e-5
            continue
        assert relmax(p.grad, g["g." + k]) < 5e-2, k
        n += 1
    assert n >= 5


def test_encoder_lstm(dev):
    """rnn_cell='lstm' (base_modules.py:42-51): (h, c) state in and out, golden from the reference's nn.LSTM path."""
    from pevae_amd.enc2dec.encoders import EncoderRNN
    from pevae_amd import ops
    g = load_golden("encoder_lstm")
    B, L, E = g["x"].shape
    H = g["h0"].shape[-1]
    enc = load_params(EncoderRNN(E, H, rnn_cell="lstm"), g).to(dev)
    x = T(g["x"], dev).requires_grad_()
    out, (hn, cn) = enc(x, init_state=(T(g["h0"], dev), T(g["c0"], dev)))
    w_hid = T(g["w_hid"], dev)
    ((out * T(g["w_out"], dev)).sum() + (hn * w_hid).sum() + (cn * w_hid).sum() * 0.5).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    close(out, g["out"], rtol=0, atol=3e-2)
    close(hn, g["hn"], rtol=0, atol=3e-2)
    close(cn, g["cn"], rtol=0, atol=3e-2)
    assert relmax(x.grad, g["dx"]) < 4e-2
    check_grads(enc, g, 4e-2)
    out0, (hn0, _) = enc(T(g["x"], dev))
    close(out0, g["out_nostate"], rtol=0, atol=3e-2)
    close(hn0, g["hn_nostate"], rtol=0, atol=3e-2)


@pytest.mark.parametrize("B,L,E,H", [(70, 6, 40, 64), (33, 4, 24, 512)])
def test_lstm_layer_vs_oracle(dev, B, L, E, H):
    """Larger LSTM shapes (several row groups / unit slices, the H=512 compile-time variant) against torch's CPU
    nn.LSTM (the ATen kernels the reference calls) with the weights rounded to bf16 as the MFMA operands are."""
    from pevae_amd import ops
    gen = torch.Generator().manual_seed(5)
    bf = lambda t: t.to(torch.bfloat16).float()
    ref = torch.nn.LSTM(E, H, batch_first=True)
    for p in ref.parameters():
        p.data = bf(torch.empty_like(p).uniform_(-0.08, 0.08, generator=gen))
    x = bf(torch.randn(B, L, E, generator=gen))
    h0, c0 = torch.randn(B, H, generator=gen) * 0.5, torch.randn(B, H, generator=gen) * 0.5
    w_o, w_h, w_c = torch.randn(B, L, H, generator=gen), torch.randn(B, H, generator=gen), torch.randn(B, H, generator=gen)
    xr, hr, cr = x.clone().requires_grad_(), h0.clone().requires_grad_(), c0.clone().requires_grad_()
    o_r, (hn_r, cn_r) = ref(xr, (hr.unsqueeze(0), cr.unsqueeze(0)))
    ((o_r * w_o).sum() + (hn_r[0] * w_h).sum() + (cn_r[0] * w_c).sum()).backward()
    prm = [p.detach().clone().to(dev).requires_grad_() for p in (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0)]
    xg, hg, cg = x.to(dev).requires_grad_(), h0.to(dev).requires_grad_(), c0.to(dev).requires_grad_()
    out, hn, cn, _ = ops.LSTMLayer.apply(xg, hg, cg, *prm, True)
    ((out * w_o.to(dev)).sum() + (hn * w_h.to(dev)).sum() + (cn * w_c.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    assert relmax(out, o_r.detach()) < 2e-2 and relmax(hn, hn_r[0].detach()) < 2e-2 and relmax(cn, cn_r[0].detach()) < 2e-2
    assert relmax(xg.grad, xr.grad) < 4e-2 and relmax(hg.grad, hr.grad) < 4e-2 and relmax(cg.grad, cr.grad) < 4e-2
    for p, q, n in zip(prm, (ref.weight_ih_l0, ref.weight_hh_l0, ref.bias_ih_l0, ref.bias_hh_l0), "w_ih w_hh b_ih b_hh".split()):
        assert relmax(p.grad, q.grad) < 4e-2, n


def test_encoder_varlen(dev):
    """variable_lengths=True (encoders.py:55-65): golden from the reference's pack_padded path; gradients against the
    oracle evaluated on the same bf16-rounded MFMA operands."""
    from pevae_amd.enc2dec.encoders import EncoderRNN
    from pevae_amd import ops
    from oracle import ref_modules as R
    g = load_golden("encoder_gru_varlen")
    enc = load_params(EncoderRNN(8, 16, variable_lengths=True), g).to(dev)
    lens = g["lens"].tolist()
    x = T(g["x"], dev).requires_grad_()
    out, hn = enc(x, input_lengths=lens)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    ((out * w.to(dev)).sum() + hn.sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    assert out.shape[1] == max(lens)
    assert relmax(out, g["out"][:, :max(lens)]) < 3e-2 and relmax(hn, g["hn"]) < 3e-2      # bf16 operands vs fp32 golden
    for b, n in enumerate(lens):
        assert float(out.detach()[b, n:].abs().max()) == 0.0 if n < out.shape[1] else True          # zero past each length
    ref = load_params(R.EncoderRNN(8, 16, variable_lengths=True), g)
    xr = T(g["x"]).requires_grad_()
    o_r, h_r = ref(xr, input_lengths=lens)
    ((o_r * w).sum() + h_r.sum()).backward()
    assert relmax(x.grad, xr.grad) < 5e-2
    for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert relmax(p.grad, q.grad) < 5e-2, n


def test_encoder_bidirectional_varlen(dev):
    """variable_lengths=True with bidirection=True: golden (outputs and gradients) from the reference's pack_padded path,
    where the reverse direction starts at each row's own last token."""
    from pevae_amd.enc2dec.encoders import EncoderRNN
    from pevae_amd import ops
    g = load_golden("encoder_gru_bi_varlen")
    enc = load_params(EncoderRNN(8, 16, variable_lengths=True, bidirection=True), g).to(dev)
    lens = g["lens"].tolist()
    x = T(g["x"], dev).requires_grad_()
    out, hn = enc(x, input_lengths=lens)
    ((out * T(g["w"], dev)).sum() + hn.sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    assert tuple(out.shape) == (5, 7, 32) and tuple(hn.shape) == (2, 5, 16)
    assert relmax(out, g["out"]) < 3e-2 and relmax(hn, g["hn"]) < 3e-2           # bf16 operands vs fp32 golden
    for b, n in enumerate(lens):
        if n < out.shape[1]:
            assert float(out.detach()[b, n:].abs().max()) == 0.0
    assert relmax(x.grad, g["dx"]) < 5e-2
    for n, p in enc.named_parameters():
        assert relmax(p.grad, g["g." + n]) < 5e-2, n


@pytest.mark.parametrize("tag,bi", [("lstm_varlen", False), ("lstm_bi_varlen", True)])
def test_encoder_lstm_varlen(dev, tag, bi):
    """variable_lengths=True with the LSTM cell (rows keep (h, c) past their length; dL/dh and dL/dc carried across the
    frozen steps): outputs, final states and gradients against goldens from the reference's pack_padded path."""
    from pevae_amd.enc2dec.encoders import EncoderRNN
    from pevae_amd import ops
    g = load_golden("encoder_" + tag)
    enc = load_params(EncoderRNN(8, 16, rnn_cell="lstm", variable_lengths=True, bidirection=bi), g).to(dev)
    lens = g["lens"].tolist()
    x = T(g["x"], dev).requires_grad_()
    out, (hn, cn) = enc(x, input_lengths=lens)
    ((out * T(g["w"], dev)).sum() + hn.sum() + 0.5 * cn.sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    assert relmax(out, g["out"]) < 3e-2 and relmax(hn, g["hn"]) < 3e-2 and relmax(cn, g["cn"]) < 3e-2
    for b, n in enumerate(lens):
        if n < out.shape[1]:
            assert float(out.detach()[b, n:].abs().max()) == 0.0
    assert relmax(x.grad, g["dx"]) < 5e-2
    for n, p in enc.named_parameters():
        assert relmax(p.grad, g["g." + n]) < 5e-2, n


def test_encoder_errors(dev):
    from pevae_amd.enc2dec.encoders import EncoderRNN
    with pytest.raises(ValueError):
        EncoderRNN(8, 16, rnn_cell="rnn")                       # base_modules.py:39
    enc = EncoderRNN(8, 16)
    with pytest.raises(RuntimeError):                            # CPU tensors: no fallback
        enc(torch.randn(2, 3, 8))


@pytest.mark.parametrize("tag,kw,lat", [("gru", {}, 0), ("gru_latent", {}, 5),
                                        ("gru_temp", dict(softmax_temperature=0.7), 0)])
def test_decoder_teacher_forcing(dev, tag, kw, lat):
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import criterions, ops
    from oracle import ref_modules as R
    g = load_golden("decoder_tf_" + tag)
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E + lat, H, 1, 2, 3, rnn_cell="gru", use_gpu=True, embedding=emb, **kw), g).to(dev)
    h0 = T(g["h0"], dev).requires_grad_()
    latent = T(g["latent"], dev) if lat else None
    outs, hid, ret = dec(B, T(g["ids"], dev), h0, mode=D.TEACH_FORCE, latent_variable=latent)
    assert isinstance(outs, D.DeferredLogProbs) and tuple(outs.size()) == (B, L, V)
    assert ret["length"] == g["length"].tolist() and ret["sequence"] == []
    # dense view == the reference's returned tensor; gradient of sum(logp * w) as in the golden
    logp = outs.materialize()
    (logp * T(g["w"], dev)).sum().backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    close(logp, g["logp"], rtol=0, atol=3e-2)
    close(hid, g["hid"], rtol=0, atol=3e-2)
    assert relmax(h0.grad, g["dh0"]) < 4e-2
    check_grads(dec, g, 4e-2)
    # fused path (no logits): NLLEntropy / Perplexity on the deferred object vs the oracle criterions on the golden
    labels = T(g["ids"])[:, 1:]
    labels = torch.cat([labels, torch.zeros(B, 1, dtype=torch.long)], 1)
    dec.zero_grad()
    outs2, _, _ = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.TEACH_FORCE, latent_variable=latent)
    nll = criterions.NLLEntropy(0, None)(outs2, labels.to(dev))
    ppl = criterions.Perplexity(0, None)(outs2, labels.to(dev))
    nllw = criterions.NLLEntropy(0, None, avg_type="word")(outs2, labels.to(dev))
    nll.backward()
    want = R.NLLEntropy(0, None)(T(g["logp"]), labels)
    assert abs(float(nll) - float(want)) < 2e-2 * abs(float(want))
    assert abs(float(ppl) - float(R.Perplexity(0, None)(T(g["logp"]), labels))) < 3e-2 * float(ppl)
    assert abs(float(torch.log(ppl)) - float(nllw)) < 1e-5
    assert outs2._dense is None                                   # the fused path never built [B,L,V]
    # gradient of the fused NLL vs autograd through the oracle decoder with the same parameters
    ref = load_params(R.DecoderRNN(V, L, E + lat, H, 1, 2, 3, rnn_cell="gru", use_gpu=False,
                                   embedding=torch.nn.Embedding(V, E, padding_idx=0), **kw), g)
    lp_ref, _, _ = ref(B, T(g["ids"]), T(g["h0"]), mode=R.TEACH_FORCE, latent_variable=T(g["latent"]) if lat else None)
    R.NLLEntropy(0, None)(lp_ref, labels).backward()
    for (k, p), (k2, p2) in zip(dec.named_parameters(), ref.named_parameters()):
        assert k == k2
        assert relmax(p.grad, p2.grad.numpy()) < 5e-2, k


def test_decoder_teacher_forcing_lstm(dev):
    """rnn_cell='lstm' through DecoderRNN (decoders.py:214-223: tuple state): golden log-probs, (h, c), gradients."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    g = load_golden("decoder_tf_lstm")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="lstm", use_gpu=True, embedding=emb), g).to(dev)
    h0, c0 = T(g["h0"], dev).requires_grad_(), T(g["c0"], dev).requires_grad_()
    outs, (hid, hid_c), ret = dec(B, T(g["ids"], dev), (h0, c0), mode=D.TEACH_FORCE)
    logp = outs.materialize()
    (logp * T(g["w"], dev)).sum().backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    close(logp, g["logp"], rtol=0, atol=3e-2)
    close(hid, g["hid"], rtol=0, atol=3e-2)
    close(hid_c, g["hid_c"], rtol=0, atol=3e-2)
    assert relmax(h0.grad, g["dh0"]) < 4e-2 and relmax(c0.grad, g["dc0"]) < 4e-2
    check_grads(dec, g, 4e-2)


@pytest.mark.parametrize("tag,mask_unk", [("greedy", False), ("greedy_maskunk", True)])
def test_decoder_gen_greedy(dev, tag, mask_unk):
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_" + tag)
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=mask_unk), g).to(dev)
    dec.eval()
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="greedy")
    seq = torch.cat(ret["sequence"], dim=1).cpu()
    want = T(g["seq"])
    # a token may legitimately flip only where the reference's top-2 log-probs are within bf16 noise
    ref_lp = T(g["logp"])
    for b in range(B):
        for t in range(L):
            if seq[b, t] != want[b, t]:
                top2 = ref_lp[b, t].topk(2)[0]
                assert float(top2[0] - top2[1]) < 3e-2, (b, t)
                break                                           # later steps diverge after a flip
        else:
            assert ret["length"][b] == int(g["length"][b])
    close(logp[:, 0], g["logp"][:, 0], rtol=0, atol=3e-2)
    with pytest.raises(ValueError):
        dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="bogus")


@pytest.mark.parametrize("mask_unk", [False, True])
def test_decoder_gen_fused_greedy(dev, mask_unk):
    """Opt-in fused arg-max decode (pv_vocab_argmax): the same symbols, lengths and per-symbol log-probabilities as the
    dense path (the reference's own return value, golden-tested above), with no [B,V] tensor."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_greedy_maskunk" if mask_unk else "decoder_gen_greedy")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=mask_unk), g).to(dev)
    dec.eval()
    with torch.no_grad():
        logp, _, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="greedy")
        dec.fused_greedy = True
        lp_f, _, ret_f = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="greedy")
    seq, seq_f = torch.cat(ret["sequence"], 1), torch.cat(ret_f["sequence"], 1)
    assert torch.equal(seq, seq_f) and ret["length"] == ret_f["length"]          # bit-exact indices
    assert tuple(lp_f.shape) == (B, L)
    close(lp_f, logp.gather(2, seq.unsqueeze(-1)).squeeze(-1).cpu(), rtol=0, atol=1e-4)


def test_decoder_gen_graphed_greedy(dev):
    """Opt-in graph_decode: the whole greedy decode replayed from one HIP graph gives what the eager fused path gives --
    for the inputs it was captured with, for new inputs, and after the weights have changed."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_greedy")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    dec.fused_greedy = True
    ids = T(g["ids"], dev)
    gen = torch.Generator().manual_seed(5)
    states = [T(g["h0"], dev), torch.randn(g["h0"].shape, generator=gen).to(dev), T(g["h0"], dev) * 0.5]
    with torch.no_grad():
        for k, h0 in enumerate(states):
            if k == 2:
                dec.project.weight.mul_(1.5)                  # a replay must see updated parameters
                dec.rnn.weight_hh_l0.add_(0.05)
            dec.graph_decode = False
            lp_e, hid_e, ret_e = dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
            dec.graph_decode = True
            lp_g, hid_g, ret_g = dec(B, ids, h0, mode=D.GEN, gen_type="greedy")
            assert torch.equal(torch.cat(ret_e["sequence"], 1), torch.cat(ret_g["sequence"], 1)), k
            assert ret_e["length"] == ret_g["length"]
            assert torch.equal(lp_e, lp_g) and torch.equal(hid_e, hid_g), k
    assert len(dec._decode_graphs) == 1


def test_decoder_gen_graphed_sample(dev):
    """graph_decode with gen_type 'sample': the whole sampling decode replayed from one HIP graph.  From the same generator
    state a replay draws exactly what the eager fused path draws (the step counter is advanced inside the graph); the next
    replay draws anew; symbols stay inside the vocabulary and a masked <unk> never appears."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    g = load_golden("decoder_gen_sample")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=True), g).to(dev)
    dec.eval()
    dec.fused_greedy = True
    ids, h0 = T(g["ids"], dev), T(g["h0"], dev)
    ops.rng_seed(20240608, dev)                       # the draws do not depend on what ran before this test
    rng = ops.rng_state(dev)
    with torch.no_grad():
        dec.graph_decode = True
        dec(B, ids, h0, mode=D.GEN, gen_type="sample")                      # warm-up + capture + first replay
        for trial in range(2):
            before = rng.clone()
            dec.graph_decode = False
            lp_e, _, ret_e = dec(B, ids, h0, mode=D.GEN, gen_type="sample")
            rng.copy_(before)
            dec.graph_decode = True
            lp_g, _, ret_g = dec(B, ids, h0, mode=D.GEN, gen_type="sample")
            seq_e, seq_g = torch.cat(ret_e["sequence"], 1), torch.cat(ret_g["sequence"], 1)
            assert torch.equal(seq_e, seq_g) and torch.equal(lp_e, lp_g) and ret_e["length"] == ret_g["length"]
            assert int(seq_g.min()) >= 0 and int(seq_g.max()) < V and not bool((seq_g == dec.unk_id).any())
        seq_next = torch.cat(dec(B, ids, h0, mode=D.GEN, gen_type="sample")[2]["sequence"], 1)
        assert not torch.equal(seq_next, seq_g)                              # the next replay: new draws
    assert len(dec._decode_graphs) == 1


@pytest.mark.parametrize("shape", ["golden", "baseline"])
def test_decoder_gen_persistent_sample(dev, shape):
    """persistent_decode with gen_type 'sample' (pv_sample_decode: the multinomial draw inside the one-launch decode): from
    the same generator state the kernel draws what the per-token fused path draws (pv_vocab_sample at call site
    SAMPLE_SITE + token) -- rows may part only where two perturbed scores nearly tie, since the two paths accumulate their
    logits in different orders --, the log-probabilities of the emitted symbols agree, a masked <unk> never appears, and the
    next decode draws anew."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    if shape == "golden":
        g = load_golden("decoder_gen_sample")
        V, L, E, H, B = 37, 7, 12, 16, 4
        emb = torch.nn.Embedding(V, E, padding_idx=0)
        dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=True), g).to(dev)
        ids, h0 = T(g["ids"], dev), T(g["h0"], dev)
    else:
        V, L, E, H, B = 20004, 50, 100, 512, 256
        torch.manual_seed(4)
        emb = torch.nn.Embedding(V, E, padding_idx=0)
        dec = D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=True).to(dev)
        with torch.no_grad():
            dec.project.weight.mul_(3.0)                  # a peaked distribution: sequences are not pure noise
        ids = torch.zeros((B, L), dtype=torch.long, device=dev)
        h0 = (torch.randn(1, B, H) * 0.5).to(dev)
    dec.eval()
    dec.fused_greedy = True
    ops.rng_seed(20240607, dev)                       # the draws do not depend on what ran before this test
    rng = ops.rng_state(dev)
    with torch.no_grad():
        before = rng.clone()
        lp_e, _, ret_e = dec(B, ids, h0, mode=D.GEN, gen_type="sample")
        rng.copy_(before)
        dec.persistent_decode = True
        lp_p, hid_p, ret_p = dec(B, ids, h0, mode=D.GEN, gen_type="sample")
        seq_next = torch.cat(dec(B, ids, h0, mode=D.GEN, gen_type="sample")[2]["sequence"], 1)
    torch.cuda.synchronize()
    ops.greedy_decode_check()
    seq_e, seq_p = torch.cat(ret_e["sequence"], 1), torch.cat(ret_p["sequence"], 1)
    same = (seq_e == seq_p).all(1)
    print("rows with the per-token path's symbols:", int(same.sum()), "of", B)
    assert int(same.sum()) >= (B if shape == "golden" else B - 6), int(same.sum())
    assert tuple(lp_p.shape) == (B, L) and int(seq_p.min()) >= 0 and int(seq_p.max()) < V
    assert not bool((seq_p == dec.unk_id).any()) and not torch.equal(seq_next, seq_p)
    close(lp_p[same], lp_e[same].cpu(), rtol=0, atol=2e-3)
    lens_e, lens_p = np.asarray(ret_e["length"]), np.asarray(ret_p["length"])
    assert (lens_e[same.cpu().numpy()] == lens_p[same.cpu().numpy()]).all()


@pytest.mark.parametrize("mask_unk", [False, True])
def test_decoder_gen_persistent_greedy_golden(dev, mask_unk):
    """Opt-in persistent_decode (pv_greedy_decode: all tokens in one launch) on the reference's own greedy run: the emitted
    symbols and lengths of the golden, the per-symbol log-probabilities and final state of the dense path."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    g = load_golden("decoder_gen_greedy_maskunk" if mask_unk else "decoder_gen_greedy")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb, mask_unk=mask_unk), g).to(dev)
    dec.eval()
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="greedy")
        dec.persistent_decode = True
        lp_p, hid_p, ret_p = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="greedy")
    torch.cuda.synchronize()
    ops.greedy_decode_check()
    seq, seq_p = torch.cat(ret["sequence"], 1), torch.cat(ret_p["sequence"], 1)
    assert torch.equal(seq, seq_p) and ret["length"] == ret_p["length"]
    assert torch.equal(seq_p.cpu(), T(g["seq"])) and ret_p["length"] == [int(v) for v in g["length"]]
    assert tuple(lp_p.shape) == (B, L) and tuple(hid_p.shape) == (1, B, H)
    close(lp_p, logp.gather(2, seq.unsqueeze(-1)).squeeze(-1).cpu(), rtol=0, atol=1e-4)
    close(hid_p, hid.cpu(), rtol=0, atol=1e-5)
    close(hid_p, g["hid"], rtol=0, atol=3e-2)


def _first_flip_is_a_near_tie(seq_a, seq_b, lp_a, margin):
    """Rows of two greedy decodes may part only where the chosen symbol of one has (almost) the log-probability of the
    other's: returns the number of rows that differ, asserting that at each row's first difference."""
    diff = seq_a != seq_b
    rows = diff.any(1).nonzero().view(-1)
    for b in rows.tolist():
        t = int(diff[b].float().argmax())
        assert abs(float(lp_a[b, t, seq_a[b, t]] - lp_a[b, t, seq_b[b, t]])) < margin, (b, t)
    return len(rows)


@pytest.mark.parametrize("B,H,V,L,E,Z", [(256, 512, 20004, 50, 100, 0), (100, 512, 20004, 12, 100, 24),
                                         (40, 96, 1000, 9, 20, 0), (70, 80, 333, 6, 12, 5)])
def test_decoder_gen_persistent_greedy_sizes(dev, B, H, V, L, E, Z):
    """pv_greedy_decode against the per-token path (dense log-probs, the reference's return value) at the BASELINE
    decoder shape (compile-time K, 256 workgroups, 80 resident columns each), with a latent input and a ragged batch,
    and at sizes that take the run-time-K kernel with zero-padded K and unit slices."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    torch.manual_seed(3)
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    # a contractive recurrence at H = 512 (uniform(-0.3, 0.3) there is chaotic: summation-order noise of 1e-7 grows to 1e-1
    # in 40 steps and the comparison would say nothing), with a projection strong enough for distinct top-2 symbols
    dec = D.DecoderRNN(V, L, E + Z, H, 2, 3, 1, rnn_cell="gru", use_gpu=True, embedding=emb,
                       init_range=0.3 if H < 128 else 0.04, mask_unk=True).to(dev)
    dec.eval()
    if H >= 128:
        with torch.no_grad():
            dec.project.weight.mul_(8.0)
    h0 = torch.randn(1, B, H, device=dev) * 0.5
    lat = torch.randn(B, Z, device=dev) if Z else None
    ids = torch.zeros(B, L, dtype=torch.long, device=dev)
    with torch.no_grad():
        logp, hid, ret = dec(B, ids, h0, mode=D.GEN, gen_type="greedy", latent_variable=lat)
        dec.persistent_decode = True
        lp_p, hid_p, ret_p = dec(B, ids, h0, mode=D.GEN, gen_type="greedy", latent_variable=lat)
        lp_p2, hid_p2, ret_p2 = dec(B, ids, h0, mode=D.GEN, gen_type="greedy", latent_variable=lat)
    torch.cuda.synchronize()
    ops.greedy_decode_check()
    seq, seq_p = torch.cat(ret["sequence"], 1), torch.cat(ret_p["sequence"], 1)
    assert torch.equal(seq_p, torch.cat(ret_p2["sequence"], 1)) and torch.equal(lp_p, lp_p2) and torch.equal(hid_p, hid_p2)
    assert int(seq_p.min()) >= 0 and int(seq_p.max()) < V and not bool((seq_p == dec.unk_id).any())
    assert len(torch.unique(seq_p)) > 3                          # not a degenerate decode
    # the two paths sum in different orders: a bf16 rounding flip of a state element moves later logits by ~1e-2 (the
    # module-level bf16 tolerance of this file)
    flipped = _first_flip_is_a_near_tie(seq, seq_p, logp, 5e-2)
    assert flipped <= max(1, B // 20), flipped
    same = (seq == seq_p).all(1)
    close(lp_p[same], logp.gather(2, seq.unsqueeze(-1)).squeeze(-1)[same].cpu(), rtol=0, atol=3e-2)
    close(hid_p[0][same], hid[0][same].cpu(), rtol=0, atol=3e-2)
    assert [ret_p["length"][b] for b in same.nonzero().view(-1).tolist()] == \
        [ret["length"][b] for b in same.nonzero().view(-1).tolist()]


def test_decoder_gen_sample_golden(dev):
    """`sample` decoding (decoders.py:255-256 -> base_modules.py:54-59) against the reference's own run: the Exp(1) draws
    behind torch.multinomial are part of the golden, so the drawn symbols can be compared token by token.  A symbol may
    differ only where the reference's two best ratios p / q are within the bf16 noise of the log-probabilities."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_sample")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    dec.sample_noise = T(g["q"], dev)
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="sample")
    seq, want = torch.cat(ret["sequence"], dim=1).cpu(), T(g["seq"])
    score = T(g["logp"]) - torch.log(T(g["q"])).transpose(0, 1)            # log(p / q) of the reference, [B, L, V]
    exact = 0
    for b in range(B):
        for t in range(L):
            if seq[b, t] != want[b, t]:
                top2 = score[b, t].topk(2)[0]
                assert float(top2[0] - top2[1]) < 3e-2, (b, t)
                break
        else:
            exact += 1
            assert ret["length"][b] == int(g["length"][b])
    assert exact >= B - 1
    close(logp[:, 0], g["logp"][:, 0], rtol=0, atol=3e-2)
    dec.sample_noise = None                                                 # generator mode: valid symbols, varying draws
    with torch.no_grad():
        a = torch.cat(dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="sample")[2]["sequence"], 1)
        b2 = torch.cat(dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="sample")[2]["sequence"], 1)
    assert int(a.min()) >= 0 and int(a.max()) < V and not torch.equal(a, b2)


def test_decoder_gen_fused_sample(dev):
    """Opt-in fused sampling step (pv_vocab_sample: the multinomial draw as arg-max of scaled logit - log q inside the
    projection GEMM, no [B,V] tensor).  (1) With the golden's Exp(1) draws: the symbols of the reference's own run, the
    per-symbol log-probabilities of the dense path.  (2) Generator mode (Philox): the empirical distribution of 16 384
    draws from one row follows softmax of that row, another call site or step draws differently, mask_unk never appears."""
    from pevae_amd.enc2dec import decoders as D
    from pevae_amd import ops
    g = load_golden("decoder_gen_sample")
    V, L, E, H, B = 37, 7, 12, 16, 4
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    dec.sample_noise = T(g["q"], dev)
    with torch.no_grad():
        logp, _, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="sample")
        dec.fused_greedy = True
        lp_f, _, ret_f = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="sample")
    seq, seq_f, want = torch.cat(ret["sequence"], 1).cpu(), torch.cat(ret_f["sequence"], 1).cpu(), T(g["seq"])
    score = T(g["logp"]) - torch.log(T(g["q"])).transpose(0, 1)            # log(p / q) of the reference, [B, L, V]
    exact = 0
    for b in range(B):
        for t in range(L):
            if seq_f[b, t] != want[b, t]:
                top2 = score[b, t].topk(2)[0]
                assert float(top2[0] - top2[1]) < 3e-2, (b, t)
                break
        else:
            exact += 1
            assert ret_f["length"][b] == int(g["length"][b])
    assert exact >= B - 1
    same = (seq == seq_f).all(1)
    assert int(same.sum()) >= B - 1 and tuple(lp_f.shape) == (B, L)
    close(lp_f[same], logp.gather(2, torch.cat(ret["sequence"], 1).unsqueeze(-1)).squeeze(-1)[same].cpu(), rtol=0, atol=1e-4)
    # generator mode on one distribution repeated 16 384 times
    torch.manual_seed(11)
    n, H2, V2 = 16384, 64, 50
    hid = (torch.randn(1, H2, device=dev) * 0.7).repeat(n, 1)
    w, bias = torch.randn(V2, H2, device=dev) * 0.3, torch.randn(V2, device=dev) * 0.2
    ids, lp = ops.vocab_sample(hid, w, bias, inv_temp=0.8, mask_col=7, site=123)
    ids2, _ = ops.vocab_sample(hid, w, bias, inv_temp=0.8, mask_col=7, site=124)
    ops.rng_advance(dev)
    ids3, _ = ops.vocab_sample(hid, w, bias, inv_temp=0.8, mask_col=7, site=123)
    assert not torch.equal(ids, ids2) and not torch.equal(ids, ids3) and not bool((ids == 7).any())
    logits = (hid[:1].to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t() + bias) * 0.8
    logsm = torch.log_softmax(logits, -1)[0]
    p = torch.softmax(logits[0].index_fill(0, torch.tensor([7], device=dev), -float("inf")), -1)   # the masked column is never drawn
    freq = torch.bincount(ids, minlength=V2).float() / n
    sigma = torch.sqrt(p * (1 - p) / n)
    assert float(((freq - p).abs() / (sigma + 1e-4)).max()) < 5.0
    close(lp, logsm[ids].cpu(), rtol=0, atol=2e-3)                            # log-probability under the UNMASKED softmax


def test_decoder_gen_beam_golden(dev):
    """`beam` decoding (decoders.py:257-270, :307-328) against the golden built from the reference's forward_step under its
    beam rule (floor-division back pointers; see oracle/make_golden.py gen_decoder_sample_beam): raw beam symbols, the
    back-tracked 1-best sequence and the lengths over all B * beam rows."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_beam")
    V, L, E, H, B, beam = 37, 7, 12, 16, 4, int(g["beam"])
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    with torch.no_grad():
        logp, hid, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="beam", beam_size=beam)
    assert logp.shape == g["logp"].shape and len(ret["length"]) == B * beam
    close(logp[:, 0], g["logp"][:, 0], rtol=0, atol=3e-2)
    seq, want = torch.cat(ret["sequence"], dim=1).cpu(), T(g["seq"])
    # beams reorder on near-ties of cumulative scores; where the whole decode agrees with the reference's log-probs to
    # bf16 noise the 1-best sequence must be the reference's
    if float((logp.cpu() - T(g["logp"])).abs().max()) < 3e-2:
        ref_lp = T(g["logp"])
        margin = min(float(ref_lp[r, t].topk(beam + 1)[0][beam - 1] - ref_lp[r, t].topk(beam + 1)[0][beam])
                     for r in range(B * beam) for t in range(L))
        if margin > 0.1:
            assert torch.equal(seq, want) and ret["length"] == g["length"].tolist()
    assert seq.shape == want.shape and float((seq == want).float().mean()) >= 0.6


def test_decoder_gen_fused_beam(dev):
    """Opt-in fused beam step (pv_vocab_topk: per row the `beam` best symbols from the projection GEMM's slab epilogue, no
    [B * beam, V] rows): the same raw beam symbols, back-tracked 1-best sequences and lengths as the dense path (which is
    checked against the reference's golden above), per-symbol log-probabilities equal to the dense rows' entries; and
    against the golden itself under the same margin rule."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_beam")
    V, L, E, H, B, beam = 37, 7, 12, 16, 4, int(g["beam"])
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    for mask_unk in (False, True):
        dec.mask_unk = mask_unk
        with torch.no_grad():
            dec.fused_greedy = False
            logp, _, ret = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="beam", beam_size=beam)
            dec.fused_greedy = True
            lp_f, _, ret_f = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="beam", beam_size=beam)
        seq, seq_f = torch.cat(ret["sequence"], 1).cpu(), torch.cat(ret_f["sequence"], 1).cpu()
        assert tuple(lp_f.shape) == (B * beam, L) and seq_f.shape == seq.shape
        # the dense path ranks fp32 rows of the same bf16 products: only near-ties of cumulative scores may reorder beams
        lp = logp.cpu()
        margin = min(float(lp[r, t].topk(beam + 1)[0][beam - 1] - lp[r, t].topk(beam + 1)[0][beam])
                     for r in range(B * beam) for t in range(L))
        if margin > 1e-3:
            assert torch.equal(seq_f, seq) and ret_f["length"] == ret["length"]
        assert float((seq_f == seq).float().mean()) >= 0.9
        if mask_unk:
            assert not bool((seq_f == dec.unk_id).any())
    want = T(g["seq"])
    dec.mask_unk = False
    with torch.no_grad():
        _, _, ret_f = dec(B, T(g["ids"], dev), T(g["h0"], dev), mode=D.GEN, gen_type="beam", beam_size=beam)
    assert float((torch.cat(ret_f["sequence"], 1).cpu() == want).float().mean()) >= 0.6


def test_decoder_gen_graphed_beam(dev):
    """graph_decode with gen_type 'beam' (fused step): the whole beam decode -- per-token top-k, candidate ranking, back
    pointers and the final back-tracking -- replayed from one HIP graph equals the eager fused decode, for the inputs it
    was captured with and for new ones."""
    from pevae_amd.enc2dec import decoders as D
    g = load_golden("decoder_gen_beam")
    V, L, E, H, B, beam = 37, 7, 12, 16, 4, int(g["beam"])
    emb = torch.nn.Embedding(V, E, padding_idx=0)
    dec = load_params(D.DecoderRNN(V, L, E, H, 1, 2, 3, rnn_cell="gru", embedding=emb), g).to(dev)
    dec.eval()
    dec.fused_greedy = True
    ids = T(g["ids"], dev)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for k, h0 in enumerate((T(g["h0"], dev), torch.randn(g["h0"].shape, generator=gen).to(dev))):
            dec.graph_decode = False
            lp_e, hid_e, ret_e = dec(B, ids, h0, mode=D.GEN, gen_type="beam", beam_size=beam)
            dec.graph_decode = True
            lp_g, hid_g, ret_g = dec(B, ids, h0, mode=D.GEN, gen_type="beam", beam_size=beam)
            assert torch.equal(torch.cat(ret_e["sequence"], 1), torch.cat(ret_g["sequence"], 1)), k
            assert ret_e["length"] == ret_g["length"] and torch.equal(lp_e, lp_g) and torch.equal(hid_e, hid_g), k
    assert len(dec._decode_graphs) == 1


def test_vocab_topk_large(dev):
    """pv_vocab_topk at the step's vocabulary (V = 20004, ragged last slab), k = 10 and 16, with and without a masked column,
    against torch.topk on the log-softmax of the same bf16 operands: same symbols in the same order up to near-ties, same
    log-probabilities; k = 1 agrees with pv_vocab_argmax."""
    from pevae_amd import ops
    gen = torch.Generator().manual_seed(10)
    Tn, H, V = 160, 512, 20004
    hid = (torch.randn(Tn, H, generator=gen) * 0.5).to(torch.bfloat16).float()
    W = torch.randn(V, H, generator=gen) * 0.05
    W[5000:5064] *= 3.0                                   # one slab holds several of a row's best entries
    W = W.to(torch.bfloat16).float()
    b = torch.randn(V, generator=gen) * 0.1
    ref = torch.log_softmax((hid.double() @ W.double().t() + b.double()) / 0.7, -1)
    for k, mask in ((10, -1), (16, 5003), (1, -1)):
        r = ref.clone()
        if mask >= 0:
            r[:, mask] = -float("inf")
        want_lp, want_id = r.topk(k, -1)
        ids, lp = ops.vocab_topk(hid.to(dev), W.to(dev), b.to(dev), k, 1.0 / 0.7, mask)
        ids, lp = ids.cpu(), lp.cpu()
        assert ids.shape == (Tn, k)
        np.testing.assert_allclose(lp.numpy(), want_lp.float().numpy(), rtol=0, atol=2e-3)
        for t in range(Tn):
            for j in range(k):                    # a flip is legitimate only between near-equal candidates
                if ids[t, j] != want_id[t, j]:
                    assert abs(float(want_lp[t, j] - r[t, ids[t, j]])) < 1e-3, (t, j)
            assert len(set(ids[t].tolist())) == k and (mask < 0 or mask not in ids[t].tolist())
        if k == 1:
            ids1, lp1 = ops.vocab_argmax(hid.to(dev), W.to(dev), b.to(dev), 1.0 / 0.7, mask)
            agree = (ids1.cpu() == ids[:, 0])
            assert float(agree.float().mean()) > 0.98
            np.testing.assert_allclose(lp1.cpu().numpy(), lp[:, 0].numpy(), rtol=0, atol=1e-4)
    # a vocabulary smaller than k slabs, and k larger than the allowed columns of a tiny vocabulary
    ids, lp = ops.vocab_topk(hid[:8, :64].contiguous().to(dev), W[:70, :64].contiguous().to(dev), b[:70].to(dev), 16, 1.0, -1)
    r = torch.log_softmax(hid[:8, :64].double() @ W[:70, :64].double().t() + b[:70].double(), -1)
    np.testing.assert_allclose(lp.cpu().numpy(), r.topk(16, -1)[0].float().numpy(), rtol=0, atol=2e-3)


def test_vocab_argmax_large(dev):
    """pv_vocab_argmax at the step's vocabulary (V = 20004, ragged last slab) against torch on the same bf16 operands."""
    from pevae_amd import ops
    gen = torch.Generator().manual_seed(9)
    Tn, H, V = 200, 512, 20004
    hid = (torch.randn(Tn, H, generator=gen) * 0.5).to(torch.bfloat16).float()
    W = (torch.randn(V, H, generator=gen) * 0.05).to(torch.bfloat16).float()
    b = torch.randn(V, generator=gen) * 0.1
    ref = torch.log_softmax((hid.double() @ W.double().t() + b.double()) / 0.7, -1)
    for mask in (-1, 3):
        r = ref.clone()
        if mask >= 0:
            r[:, mask] = -float("inf")
        want_lp, want_id = r.max(-1)
        ids, lp = ops.vocab_argmax(hid.to(dev), W.to(dev), b.to(dev), 1.0 / 0.7, mask)
        ids, lp = ids.cpu(), lp.cpu()
        for t in range(Tn):                      # a flip is legitimate only between near-equal candidates
            if ids[t] != want_id[t]:
                assert float(want_lp[t] - r[t, ids[t]]) < 1e-3, t
        close(lp, want_lp.float(), rtol=0, atol=2e-3)
        assert int(ids.min()) >= 0 and (mask < 0 or int((ids == mask).sum()) == 0)


@pytest.mark.parametrize("mode", ["dot", "general", "cat"])
def test_attention(dev, mode):
    from pevae_amd.enc2dec.decoders import Attention
    g = load_golden("attention_" + mode)
    att = load_params(Attention(16, 16, mode, project=True), g).to(dev)
    out, ctx = T(g["out"], dev).requires_grad_(), T(g["ctx"], dev).requires_grad_()
    y, a = att(out, ctx)
    (y * T(g["w"], dev)).sum().backward()
    close(y, g["y"]); close(a, g["attn"])
    close(out.grad, g["dout"], rtol=1e-4, atol=2e-5); close(ctx.grad, g["dctx"], rtol=1e-4, atol=2e-5)
    for k, p in att.named_parameters():
        close(p.grad, g["g." + k], rtol=2e-4, atol=2e-5)
    with pytest.raises(ValueError):
        Attention(16, 16, "nope").to(dev)(out, ctx)


def test_connectors(dev):
    from pevae_amd import nn_lib
    g = load_golden("hidden2gaussian")
    m = load_params(nn_lib.Hidden2Gaussian(16, 8), g).to(dev)
    x = T(g["x"], dev).requires_grad_()
    mu, lv = m(x)
    ((mu * T(g["wm"], dev)).sum() + (lv * T(g["wl"], dev)).sum()).backward()
    close(mu, g["mu"]); close(lv, g["logvar"]); close(x.grad, g["dx"])
    for k, p in m.named_parameters():
        close(p.grad, g["g." + k])
    g = load_golden("hidden2gaussian_lstm")
    m = load_params(nn_lib.Hidden2Gaussian(16, 8, is_lstm=True), g).to(dev)
    mu, lv = m((T(g["h"], dev), T(g["c"], dev)))
    close(mu, g["mu"]); close(lv, g["logvar"])
    g = load_golden("linear_connector")
    m = load_params(nn_lib.LinearConnector(8, 16, False), g).to(dev)
    z = T(g["z"], dev).requires_grad_()
    y = m(z)
    (y * T(g["w"], dev)).sum().backward()
    close(y, g["y"]); close(z.grad, g["dz"])
    g = load_golden("linear_connector_lstm")
    m = load_params(nn_lib.LinearConnector(8, 16, True), g).to(dev)
    yh, yc = m(T(g["z"], dev))
    close(yh, g["yh"]); close(yc, g["yc"])
    g = load_golden("greedy_connector")
    y, ids = nn_lib.GreedyConnector()(T(g["logits"], dev), True, return_max_id=True)
    assert torch.equal(ids.cpu(), T(g["ids"])); close(y, g["y"])


@pytest.mark.parametrize("tag,kw", [("soft", {}), ("hard", dict(hard=True, return_max_id=True)),
                                    ("st", dict(straight_through=True))])
def test_gumbel_connector(dev, tag, kw):
    from pevae_amd import nn_lib
    g = load_golden("gumbel_" + tag)
    logits = T(g["logits"], dev).requires_grad_()
    res = nn_lib.GumbelConnector()(logits, float(g["tau"]), True, u=T(g["u"], dev), **kw)
    y, ids = res if isinstance(res, tuple) else (res, None)
    (y * T(g["w"], dev)).sum().backward()
    close(y, g["y"], rtol=1e-5, atol=2e-6)
    close(logits.grad, g["dlogits"], rtol=1e-4, atol=2e-6)
    if ids is not None:
        assert torch.equal(ids.cpu(), T(g["ids"]))             # arg-max ids: bit-exact
    torch.manual_seed(42)                                       # the reference's own CPU draw (nn_lib.py:231)
    res2 = nn_lib.GumbelConnector()(T(g["logits"], dev), float(g["tau"]), True, **kw)
    close(res2[0] if isinstance(res2, tuple) else res2, g["y"], rtol=1e-5, atol=2e-6)


def test_criterions_dense_and_kl(dev):
    from pevae_amd import criterions as C
    g = load_golden("criterions_nll")
    labels = T(g["labels"], dev)
    for avg in ("seq", "word", None):
        logp = T(g["logp"], dev).requires_grad_()
        v = C.NLLEntropy(0, None, avg_type=avg)(logp, labels)
        v.backward()
        close(v, g["nll_%s" % avg], rtol=1e-5); close(logp.grad, g["dlogp_%s" % avg], rtol=1e-5, atol=1e-7)
    logp = T(g["logp"], dev).requires_grad_()
    v = C.Perplexity(0, None)(logp, labels)
    v.backward()
    close(v, g["ppl"], rtol=1e-5); close(logp.grad, g["dlogp_ppl"], rtol=1e-4, atol=1e-6)
    with pytest.raises(NotImplementedError):
        C.NLLEntropy(0, None, avg_type="bogus")                 # criterions.py:52

    g = load_golden("criterions_normkl")
    for ua in (False, True):
        ts = [T(g[k], dev).requires_grad_() for k in ("mq", "lq", "mp", "lp")]
        v = C.NormKLLoss(unit_average=ua)(ts[0], ts[1] * 0.5, ts[2], ts[3] * 0.5)
        v.backward()
        close(v, g["kl_ua%d" % ua], rtol=2e-5)
        for n, t in zip(("mq", "lq", "mp", "lp"), ts):
            close(t.grad, g["d%s_ua%d" % (n, ua)], rtol=1e-4, atol=1e-6)
    same = C.NormKLLoss()(T(g["mq"], dev), T(g["lq"], dev), T(g["mq"], dev), T(g["lq"], dev))
    assert abs(float(same)) < 1e-6

    g = load_golden("criterions_cat")
    lpy = T(g["log_py"], dev)
    ck = C.CatKLLoss()
    for tag, kw in (("bs", dict(batch_size=4)), ("ua", dict(unit_average=True)), ("rows", dict(average=False))):
        lq = T(g["log_qy"], dev).requires_grad_()
        v = ck(lq, lpy, **kw)
        v.sum().backward()
        close(v, g["catkl_" + tag], rtol=1e-5, atol=1e-6); close(lq.grad, g["dlq_" + tag], rtol=1e-5, atol=1e-6)
    lq = T(g["log_qy"], dev)
    close(ck(lq, torch.log(torch.ones(1) / 5).to(dev), batch_size=4), g["catkl_uniform"], rtol=1e-5, atol=1e-6)
    close(ck(lq.view(1, 12, 5), lpy, batch_size=4), g["catkl_3d"], rtol=1e-5, atol=1e-6)
    assert abs(float(ck(lq, lq, batch_size=4))) < 1e-6
    close(C.Entropy()(lq, batch_size=4), g["ent_bs"], rtol=1e-5)
    close(C.Entropy()(lq, unit_average=True), g["ent_ua"], rtol=1e-5)
