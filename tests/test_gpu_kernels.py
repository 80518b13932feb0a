"""GPU parity tests, kernel level: every call goes through the C ABI (ctypes) and is checked against the CPU
oracle (oracle/ref_modules.py) or an exact integer / sequential-order reference on the same seeded inputs.

Tolerances (stated per test): integer / index / byte work is bit-exact; fp32 kernels 1e-5; kernels that round
operands to bf16 for the MFMA are compared (a) tightly against the oracle evaluated on the SAME bf16-rounded
operands and (b) loosely against the pure-fp32 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_modules as R


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pevae_amd import ops
    info = ops.device_info()
    assert "gfx950" in info["arch"], info
    return torch.device("cuda:0")


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_gemm_exact_integers(dev):
    """A = small integers (exact in bf16), asymmetric B: catches row/col swaps and fragment-map mistakes."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(128, 128, 64), (200, 300, 128), (77, 513, 192), (1, 5, 64)]:
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        Bm = (torch.randint(-3, 4, (N, K), generator=g) + torch.arange(N).view(-1, 1) % 3).float()
        bias = torch.randint(-5, 6, (N,), generator=g).float()
        want = A @ Bm.t() + bias
        out = ops.gemm_nt(A.to(dev).to(torch.bfloat16), K, Bm.to(dev).to(torch.bfloat16), K, M, N, K,
                          bias=bias.to(dev))
        outb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ops.gemm_nt(A.to(dev).to(torch.bfloat16), K, Bm.to(dev).to(torch.bfloat16), K, M, N, K, out_bf16=outb)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), want), (M, N, K)
        assert torch.equal(outb.float().cpu(), bf(A @ Bm.t())), (M, N, K)
    # A = I with asymmetric B
    K = 64
    A = torch.eye(K)
    Bm = torch.arange(K * K).view(K, K).float() % 17
    out = ops.gemm_nt(A.to(dev).to(torch.bfloat16), K, Bm.to(dev).to(torch.bfloat16), K, K, K, K)
    assert torch.equal(out.cpu(), Bm.t().contiguous())


@pytest.mark.parametrize("T,V,H,temp", [(300, 1000, 64, 1.0), (130, 37, 16, 0.7), (64, 20004, 128, 1.0)])
def test_vocab_nll_fused(dev, T, V, H, temp):
    from pevae_amd import ops
    torch.manual_seed(2)
    hid = torch.randn(T, H) * 0.7
    W = torch.randn(V, H) * 0.2
    b = torch.randn(V) * 0.1
    tgt = torch.randint(1, V, (T,))
    tgt[::7] = 0                                     # ignored (pad) tokens
    wtok = torch.rand(T) + 0.5
    # oracle on bf16-rounded operands (what the MFMA sees), fp32 math
    hr, Wr = bf(hid).requires_grad_(), bf(W).requires_grad_()
    br = b.clone().requires_grad_()
    logp = R.log_softmax_rows((hr @ Wr.t() + br) / temp)
    nll_ref = -(logp.gather(1, tgt.unsqueeze(1)).squeeze(1)) * (tgt != 0)
    (nll_ref * wtok).sum().backward()
    hd, Wd, bd = hid.to(dev).requires_grad_(), W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    nll, lse = ops.VocabNLL.apply(hd, Wd, bd, tgt.to(dev), 1.0 / temp, 0, None, None)
    (nll * wtok.to(dev)).sum().backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(nll.detach().cpu().numpy(), nll_ref.detach().numpy(), rtol=2e-4, atol=2e-4)
    lse_ref = torch.logsumexp((hr @ Wr.t() + br) / temp, dim=1)
    # rows of ignored tokens never reach the GEMMs (ops.SKIP_IGNORED_TOKENS): their lse is reported as 0
    live = (tgt != 0) if ops.SKIP_IGNORED_TOKENS else torch.ones_like(tgt, dtype=torch.bool)
    np.testing.assert_allclose(lse.cpu()[live].numpy(), lse_ref.detach()[live].numpy(), rtol=2e-4, atol=2e-4)
    if ops.SKIP_IGNORED_TOKENS:
        assert float(lse.cpu()[~live].abs().max()) == 0.0
    # gradients: P is rounded to bf16 before the two gradient GEMMs -> 1e-2 of the largest entry
    assert relerr(hd.grad, hr.grad) < 1.5e-2
    assert relerr(Wd.grad, Wr.grad) < 1.5e-2
    assert relerr(bd.grad, br.grad) < 1.5e-2
    # dense path: same numbers as a tensor
    dense = ops.VocabLogProbDense.apply(hid.to(dev), W.to(dev), b.to(dev), 1.0 / temp)
    np.testing.assert_allclose(dense.cpu().numpy(), logp.detach().numpy(), rtol=2e-4, atol=3e-4)
    # pure fp32 oracle, loose: bf16 operand rounding only
    logp32 = R.log_softmax_rows((hid @ W.t() + b) / temp)
    nll32 = -(logp32.gather(1, tgt.unsqueeze(1)).squeeze(1)) * (tgt != 0)
    assert float((nll.detach().cpu() - nll32).abs().max()) < 0.05


@pytest.mark.parametrize("T,V,temp,case", [(300, 2003, 1.0, "plain"), (1000, 5000, 0.7, "plain"), (130, 1999, 1.0, "spike"),
                                           (257, 700, 1.0, "dense"), (64, 640, 1.0, "one_live"), (96, 333, 1.0, "all_ignored")])
def test_vocab_fused_kernels(dev, T, V, temp, case):
    """csrc/vocab_fused.hip (H = 512: S^T = W h^T -> online softmax -> O^T += W^T P^T, and S -> P -> Z += P^T h, nothing of
    size [T, V] in HBM) through ops.VocabNLL against the oracle on the same bf16-rounded operands: nll, lse, dH, dW, dbias.
    Cases: ragged vocabularies / row counts (partial 32-column and 32-row items, several stream-K segments per unit), a
    logit spike late in the vocabulary that forces the rescale branch of the online softmax (rows whose maximum grows by
    > 20 after the first tile), no compaction (every row through the kernels), a single live row, no live row at all;
    and a second run must reproduce every bit."""
    from pevae_amd import _lib, ops
    H = 512
    assert _lib.load().pv_vocab_fused_supported(H) == 1 and ops.FUSED_VOCAB
    g = torch.Generator().manual_seed(T + V)
    hid = torch.randn(T, H, generator=g) * 0.5
    W = torch.randn(V, H, generator=g) * 0.05
    b = torch.randn(V, generator=g) * 0.1
    tgt = torch.randint(1, V, (T,), generator=g)
    tgt[::3] = 0
    if case == "spike":                       # columns past the first tiles dominate some rows by ~ 60 nats
        W[V - 40] = hid[5] * 0.5
        W[V - 300] = hid[77] * 0.4
        b[V - 7] = 45.0
        tgt[5], tgt[77] = V - 40, 11
    if case == "one_live":
        tgt[:] = 0
        tgt[37] = 5
    if case == "all_ignored":
        tgt[:] = 0
    wtok = torch.rand(T, generator=g) + 0.5
    hr, Wr = bf(hid).requires_grad_(), bf(W).requires_grad_()
    br = b.clone().requires_grad_()
    logits = (hr.double() @ Wr.double().t() + br.double()) / temp
    logp = torch.log_softmax(logits, 1)
    nll_ref = -(logp.gather(1, tgt.unsqueeze(1)).squeeze(1)) * (tgt != 0)
    (nll_ref * wtok.double()).sum().backward()
    lse_ref = torch.logsumexp(logits, 1).float().detach()
    skip = ops.SKIP_IGNORED_TOKENS
    ops.SKIP_IGNORED_TOKENS = case != "dense"
    try:
        outs = []
        for _ in range(2):
            hd, Wd, bd = hid.to(dev).requires_grad_(), W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
            nll, lse = ops.VocabNLL.apply(hd, Wd, bd, tgt.to(dev), 1.0 / temp, 0, None, None)
            (nll * wtok.to(dev)).sum().backward()
            torch.cuda.synchronize()
            outs.append((nll.detach().clone(), lse.clone(), hd.grad.clone(), Wd.grad.clone(), bd.grad.clone()))
    finally:
        ops.SKIP_IGNORED_TOKENS = skip
    nll, lse, dh, dw, db = outs[0]
    np.testing.assert_allclose(nll.cpu().numpy(), nll_ref.detach().float().numpy(), rtol=3e-4, atol=3e-4)
    live = (tgt != 0) if case != "dense" else torch.ones_like(tgt, dtype=torch.bool)
    if bool(live.any()):
        np.testing.assert_allclose(lse.cpu()[live].numpy(), lse_ref[live].numpy(), rtol=3e-4, atol=3e-4)
    assert float(lse.cpu()[~live].abs().max() if bool((~live).any()) else 0.0) == 0.0
    if case == "all_ignored":
        assert not dh.any() and not dw.any() and not db.any() and not nll.any()
    else:
        # probabilities are rounded to bf16 between the two products: 1e-2 of the largest entry
        assert relerr(dh, hr.grad.float()) < 1.5e-2
        assert relerr(dw, Wr.grad.float()) < 1.5e-2
        assert relerr(db, br.grad.float()) < 1.5e-2
        assert not dh.cpu()[tgt == 0].any()               # ignored rows: exactly zero gradient
    for a, c in zip(outs[0], outs[1]):
        assert torch.equal(a, c)                          # fixed-order folds, no atomics: bitwise reproducible
    # the tile-GEMM path (P / P^T through HBM) on the same inputs agrees to the same tolerance
    ops.FUSED_VOCAB = False
    try:
        hd, Wd, bd = hid.to(dev).requires_grad_(), W.to(dev).requires_grad_(), b.to(dev).requires_grad_()
        nll2, _ = ops.VocabNLL.apply(hd, Wd, bd, tgt.to(dev), 1.0 / temp, 0, None, None)
        (nll2 * wtok.to(dev)).sum().backward()
    finally:
        ops.FUSED_VOCAB = True
    np.testing.assert_allclose(nll.detach().cpu().numpy(), nll2.detach().cpu().numpy(), rtol=3e-4, atol=3e-4)
    if case != "all_ignored":
        assert relerr(dw, Wd.grad) < 2e-2 and relerr(dh, hd.grad) < 2e-2


def test_vocab_fused_dw_on_fewer_workgroups_into_sinks(dev):
    """pv_vocab_fused_dw_wgs: the stream-K split of the weight-gradient kernel takes any workgroup count, and dW / dbias
    accumulate into existing gradients (the flat buffer's views).  100 workgroups + accumulation against one-per-CU +
    returned gradients: the same numbers up to the order of the segment folds; and the deferred launch
    (ops.defer_after_recurrence, here flushed by join_side_work since no recurrence follows) leaves the same bits as the
    immediate one."""
    from pevae_amd import ops
    T, V, H = 700, 3001, 512
    g = torch.Generator().manual_seed(5)
    hid = (torch.randn(T, H, generator=g) * 0.5).to(dev)
    W0 = (torch.randn(V, H, generator=g) * 0.05).to(dev)
    b0 = (torch.randn(V, generator=g) * 0.1).to(dev)
    tgt = torch.randint(1, V, (T,), generator=g)
    tgt[::4] = 0
    tgt = tgt.to(dev)

    def run(sinks, wgs, deferred):
        W, b = torch.nn.Parameter(W0.clone()), torch.nn.Parameter(b0.clone())
        if sinks:
            W.grad, b.grad = torch.full_like(W, 0.5), torch.full_like(b, -0.25)
        hd = hid.clone().requires_grad_()
        prev = ops.VOCAB_DW_SIDE_WGS, ops.ASYNC_WGRAD
        ops.VOCAB_DW_SIDE_WGS, ops.ASYNC_WGRAD = wgs, deferred
        try:
            nll, _ = ops.VocabNLL.apply(hd, W, b, tgt, 1.0, 0, None, None)
            nll.sum().backward()
        finally:
            ops.VOCAB_DW_SIDE_WGS, ops.ASYNC_WGRAD = prev
            ops.join_side_work()
        torch.cuda.synchronize()
        return W.grad.clone(), b.grad.clone(), hd.grad.clone()

    dw_a, db_a, dh_a = run(False, 0, False)
    dw_b, db_b, dh_b = run(True, 100, True)
    assert torch.equal(dh_a, dh_b)
    assert relerr(dw_b - 0.5, dw_a) < 1e-5 and relerr(db_b + 0.25, db_a) < 1e-5
    dw_c, db_c, _ = run(True, 100, True)
    assert torch.equal(dw_b, dw_c) and torch.equal(db_b, db_c)


@pytest.mark.parametrize("fractions", [(0.5, 0.5), (0.3, 0.3)])
def test_vocab_dw_in_row_ranges_beside_recurrences(dev, fractions):
    """ops.VOCAB_DW_BESIDE / pv_vocab_fused_dw_range: the weight gradient of the vocabulary projection in two (or three) row
    ranges -- launched where a backward recurrence would start (ops.launch_beside_recurrence: second side stream, half the
    chip) or, if none comes, at the end of the backward pass on the whole chip -- accumulates into the sinks what the single
    launch does, up to the order of the stream-K segment folds, bit-identically from run to run; targets of every range hit
    their one-hot rows (the bias gradient sums to zero over the vocabulary)."""
    from pevae_amd import ops
    T, V, H = 900, 5003, 512
    g = torch.Generator().manual_seed(6)
    hid = (torch.randn(T, H, generator=g) * 0.5).to(dev)
    W0 = (torch.randn(V, H, generator=g) * 0.05).to(dev)
    b0 = (torch.randn(V, generator=g) * 0.1).to(dev)
    tgt = torch.randint(1, V, (T,), generator=g)
    tgt[::5] = 0
    tgt = tgt.to(dev)

    def run(beside, recurrences):
        W, b = torch.nn.Parameter(W0.clone()), torch.nn.Parameter(b0.clone())
        W.grad, b.grad = torch.full_like(W, 0.5), torch.full_like(b, -0.25)
        hd = hid.clone().requires_grad_()
        prev = ops.VOCAB_DW_BESIDE, ops.ASYNC_WGRAD
        ops.VOCAB_DW_BESIDE, ops.ASYNC_WGRAD = beside, True
        budgets = []
        try:
            nll, _ = ops.VocabNLL.apply(hd, W, b, tgt, 1.0, 0, None, None)
            nll.sum().backward()
            for _ in range(recurrences):                  # what GRULayer.backward does in front of its launch
                ops.join_side_work()
                budgets.append(ops.launch_beside_recurrence())
        finally:
            ops.VOCAB_DW_BESIDE, ops.ASYNC_WGRAD = prev
            ops.join_side_work()                          # the end of the backward pass: what still waits runs here
        torch.cuda.synchronize()
        return W.grad.clone(), b.grad.clone(), hd.grad.clone(), budgets

    dw_a, db_a, dh_a, _ = run(None, 0)
    half = torch.cuda.get_device_properties(dev).multi_processor_count // 2
    for recurrences in (2, 1, 0):
        dw_b, db_b, dh_b, budgets = run(fractions, recurrences)
        assert budgets == [half] * recurrences
        assert torch.equal(dh_a, dh_b)
        assert relerr(dw_b - 0.5, dw_a - 0.5) < 1e-5 and relerr(db_b + 0.25, db_a + 0.25) < 1e-5
        assert abs(float((db_b + 0.25).sum())) < 1e-3
        dw_c, db_c, _, _ = run(fractions, recurrences)
        assert torch.equal(dw_b, dw_c) and torch.equal(db_b, db_c)
    assert ops.launch_beside_recurrence() == 0            # nothing is left waiting


def gru_ref(x, h0, w_ih, w_hh, b_ih, b_hh, emulate_bf16):
    """oracle GRU (R.gru_cell arithmetic) with optional bf16 rounding of the MFMA operands (straight-through)."""
    def q(t):
        return t + (bf(t) - t).detach() if emulate_bf16 else t
    B, L, _ = x.shape
    H = w_hh.shape[1]
    h = h0
    outs = []
    xp = q(x) @ q(w_ih).t() + b_ih
    for t in range(L):
        gh = q(h) @ q(w_hh).t() + b_hh
        r = torch.sigmoid(xp[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(xp[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(xp[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs.append(h)
    return torch.stack(outs, 1), h


# (6, 12, 20, 96) and (3, 9, 10, 48): several unit slices per row group with fewer state elements than sync words -- the
# arrival flags of the launch must be zeroed all the same (a recycled sync buffer holds the previous launch's step counts)
# (544, 3, 16, 512): more row groups than co-resident workgroups in both kernels (a workgroup walks several groups per step)
@pytest.mark.parametrize("B,L,E,H", [(4, 7, 12, 16), (37, 5, 20, 32), (6, 12, 20, 96), (3, 9, 10, 48), (64, 6, 100, 128),
                                     (256, 4, 100, 512), (544, 3, 16, 512)])
def test_gru_layer(dev, B, L, E, H):
    from pevae_amd import ops
    torch.manual_seed(3)
    s = 1.0 / np.sqrt(H)
    x = torch.randn(B, L, E)
    h0 = torch.randn(B, H) * 0.5
    w_ih, w_hh = (torch.rand(3 * H, E) * 2 - 1) * s, (torch.rand(3 * H, H) * 2 - 1) * s
    b_ih, b_hh = (torch.rand(3 * H) * 2 - 1) * s, (torch.rand(3 * H) * 2 - 1) * s
    w_out, w_hn = torch.randn(B, L, H), torch.randn(B, H)
    cpu = [t.clone().requires_grad_() for t in (x, h0, w_ih, w_hh, b_ih, b_hh)]
    out_r, hn_r = gru_ref(*cpu, emulate_bf16=True)
    ((out_r * w_out).sum() + (hn_r * w_hn).sum()).backward()
    gpu = [t.to(dev).requires_grad_() for t in (x, h0, w_ih, w_hh, b_ih, b_hh)]
    out, hn, hseq = ops.GRULayer.apply(*gpu, True)
    ((out * w_out.to(dev)).sum() + (hn * w_hn.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    ops.gru_check_all()
    # (a) vs the oracle on bf16-rounded MFMA operands: only summation order differs
    np.testing.assert_allclose(out.detach().cpu().numpy(), out_r.detach().numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), hn_r.detach().numpy(), rtol=0, atol=2e-3)
    # (b) vs the pure fp32 oracle (R.run_rnn through EncoderRNN arithmetic)
    out32, hn32 = gru_ref(x, h0, w_ih, w_hh, b_ih, b_hh, emulate_bf16=False)
    assert float((out.detach().cpu() - out32).abs().max()) < 3e-2
    # state sequence handed to downstream GEMMs is the bf16 image of `out`, time-major
    Hk, K3, Bpad, NG = ops.gru_geometry(B, H)
    hs = hseq.float().cpu().view(L + 1, Bpad, Hk)
    assert torch.equal(hs[1:, :B, :H], bf(out.detach().cpu()).transpose(0, 1))
    assert torch.equal(hs[0, :B, :H], bf(h0))
    # gradients: gate gradients are rounded to bf16 for the MFMA GEMMs
    names = ("x", "h0", "w_ih", "w_hh", "b_ih", "b_hh")
    for n, a, b in zip(names, gpu, cpu):
        assert relerr(a.grad, b.grad) < 2.5e-2, n


def test_gru_no_h0_and_nograd(dev):
    from pevae_amd import ops
    torch.manual_seed(4)
    B, L, E, H = 5, 3, 8, 16
    x = torch.randn(B, L, E)
    w = [torch.randn(3 * H, E) * .2, torch.randn(3 * H, H) * .2, torch.randn(3 * H) * .1, torch.randn(3 * H) * .1]
    with torch.no_grad():
        out, hn, _ = ops.GRULayer.apply(x.to(dev), None, *[t.to(dev) for t in w], False)
    out_r, hn_r = gru_ref(x, torch.zeros(B, H), *w, emulate_bf16=True)
    ops.gru_check_all()
    np.testing.assert_allclose(out.cpu().numpy(), out_r.numpy(), atol=2e-3, rtol=0)


def test_embedding_gather_scatter_bit_exact(dev):
    from pevae_amd import ops
    g = torch.Generator().manual_seed(5)
    for rows, D, shape in [(50, 100, (16, 9)), (7, 256, (33,)), (20004, 100, (8, 50))]:
        table = torch.randn(rows, D, generator=g)
        ids = torch.randint(0, rows, shape, generator=g)
        ids.view(-1)[::3] = ids.view(-1)[0]                # heavy duplicates
        ids.view(-1)[1::5] = 0                             # padding rows
        grad = torch.randn(*shape, D, generator=g)
        tw = table.to(dev).requires_grad_()
        out = ops.embedding(tw, ids.to(dev), 0)
        out.backward(grad.to(dev))
        torch.cuda.synchronize()
        assert torch.equal(out.detach().cpu(), table[ids])                    # gather: bit-exact
        want = np.zeros((rows, D), dtype=np.float32)
        gf = grad.view(-1, D).numpy()
        for r, i in enumerate(ids.view(-1).tolist()):                        # occurrence order, fp32
            if i != 0:
                want[i] += gf[r]
        assert np.array_equal(tw.grad.cpu().numpy(), want)                   # scatter-add: bit-exact
        tw.grad = None
        out = ops.embedding(tw, ids.to(dev), 0)
        out.backward(grad.to(dev))
        assert np.array_equal(tw.grad.cpu().numpy(), want)                   # and run-to-run identical
    # empty input
    e = ops.embedding(table.to(dev), torch.zeros((0, 4), dtype=torch.long, device=dev), 0)
    assert e.shape == (0, 4, D)


def test_embedding_scatter_long_segments(dev):
    """Rows with >= 256 occurrences are summed in pieces (embedding.hip SEG_CH): the head piece up to the first multiple
    of 256 that lies >= 256 past the segment head, then 256-position pieces, combined in position order.  Bit-exact
    against that association in numpy, within fp32 rounding of the sequential sum, identical run to run."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(6)
    rows, D, n, CH = 300, 100, 3000, 256
    ids = torch.randint(1, rows, (n,), generator=g)
    ids[torch.rand(n, generator=g) < 0.5] = 7               # ~1500 occurrences
    ids[torch.rand(n, generator=g) < 0.12] = 250            # ~300: crosses one or two cuts
    grad = torch.randn(n, D, generator=g)
    tw = torch.zeros(rows, D, device=dev).requires_grad_()
    ops.embedding(tw, ids.to(dev), 0).backward(grad.to(dev))
    got = tw.grad.cpu().numpy().copy()
    order = np.argsort(ids.numpy(), kind="stable")
    keys, gf = ids.numpy()[order], grad.numpy()[order]
    want = np.zeros((rows, D), dtype=np.float32)
    i = 0
    while i < n:
        j = i
        while j < n and keys[j] == keys[i]:
            j += 1
        cuts = [i] + [q for q in range(0, n, CH) if i + CH <= q < j] + [j]
        for a, b in zip(cuts[:-1], cuts[1:]):
            piece = np.zeros(D, dtype=np.float32)
            for r in range(a, b):
                piece += gf[r]
            want[keys[i]] = piece if a == i else want[keys[i]] + piece
        i = j
    assert np.array_equal(got, want)
    seq = np.zeros((rows, D), dtype=np.float64)
    np.add.at(seq, ids.numpy(), grad.numpy().astype(np.float64))
    np.testing.assert_allclose(got, seq, rtol=1e-5, atol=1e-4)
    tw.grad = None
    ops.embedding(tw, ids.to(dev), 0).backward(grad.to(dev))
    assert np.array_equal(tw.grad.cpu().numpy(), got)


def test_embedding_forward_time_sort_gives_the_same_bits(dev):
    """ops.PRESORT: the (row id, occurrence) sort made at forward time on a background stream (pv_embedding_sort) and the
    backward left with pv_embedding_bwd_sorted accumulate into a gradient sink exactly what the one-call backward does."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(8)
    rows, D = 5000, 100
    ids = torch.randint(0, rows, (64, 50), generator=g)
    ids[:, ::3] = 17
    grad = torch.randn(64, 50, D, generator=g).to(dev)
    got = []
    for presort in (False, True):
        w = torch.nn.Parameter(torch.randn(rows, D, generator=torch.Generator().manual_seed(1)).to(dev))
        w.grad = torch.full((rows, D), 0.25, device=dev)              # a sink with earlier contents: the sum accumulates
        prev, ops.PRESORT = ops.PRESORT, presort
        try:
            out = ops.embedding(w, ids.to(dev), 0)
            assert (out.grad_fn.presorted is not None) == presort
            out.backward(grad)
        finally:
            ops.PRESORT = prev
        torch.cuda.synchronize()
        got.append(w.grad.clone())
    assert torch.equal(got[0], got[1]) and float((got[0] - 0.25).abs().max()) > 0


@pytest.mark.parametrize("rows,n,n1,n1p", [(262143, 16384, 64, 64), (200001, 12800, 50, 50), (20004, 16385, 16385, 16385),
                                            (262144, 4096, 16, 32), (30000, 256, 256, 256), (5, 1, 1, 1), (1000, 777, 7, 9)])
def test_embedding_backward_sort_limits(dev, rows, n, n1, n1p):
    """pv_embedding_bwd through the C ABI at the edges of its own single-workgroup LDS sort (embedding.hip: n <= 16 384
    occurrences, tables below 2^18 rows) and just beyond them (the library sort): bit-exact against numpy's sequential
    scatter-add in occurrence order, ids outside the table ignored, padded gradient-row layout (n1p > n1)."""
    import ctypes as C
    from pevae_amd import _lib
    g = torch.Generator().manual_seed(rows + n)
    D = 8
    ids = torch.randint(0, rows, (n,), generator=g)
    ids[::97] = ids[0]                                  # duplicates (below the 256 of a cut segment)
    if n > 4:
        ids[3] = rows + 5                                   # outside the table: no gradient
        ids[4] = -2
    n0 = (n + n1 - 1) // n1
    assert n0 * n1 == n
    grad = torch.randn(n0 * n1p, D, generator=g)
    want = np.zeros((rows, D), dtype=np.float32)
    cnt = np.zeros(rows, dtype=np.int64)
    gf = grad.numpy()
    for r, i in enumerate(ids.tolist()):
        if 0 <= i < rows and i != 0:
            want[i] += gf[(r // n1) * n1p + r % n1]
            cnt[i] += 1
    assert cnt.max() < 256 or rows == 5                     # (long segments have their own test)
    ids_d, grad_d = ids.to(dev), grad.to(dev)
    dt = torch.empty(rows, D, device=dev)
    ws_bytes = _lib.load().pv_embedding_bwd_ws_bytes(n, D)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        _lib.call("pv_embedding_bwd", C.c_void_p(grad_d.data_ptr()), D, C.c_void_p(ids_d.data_ptr()), n, n1, n1p, n1, 1,
                  rows, D, 0, 0, C.c_void_p(dt.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes, st)
        torch.cuda.synchronize()
        got = dt.cpu().numpy()
        if rows == 5:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
        else:
            assert np.array_equal(got, want)


def test_reparam_kl(dev):
    from pevae_amd import ops
    torch.manual_seed(6)
    B, Z = 37, 24
    for mode in (0, 1):
        cpu = [torch.randn(B, Z).requires_grad_() for _ in range(4)]
        eps = torch.rand(B, Z)
        wz, wk = torch.randn(B, Z), torch.randn(B)
        mq, lq, mp, lp = cpu
        z_r = R.reparameterization(mq, lq * 0.3, eps)
        if mode == 0:
            t = 1.0 + (lq * 0.3 - lp * 0.3) - (mp - mq) ** 2 / torch.exp(lp * 0.3) - torch.exp(lq * 0.3) / torch.exp(lp * 0.3)
            kl_r = -0.5 * t.sum(1)
        else:
            l1, l2 = lq * 0.3, lp * 0.3
            kl_r = (-0.5 + l2 - l1 + 0.5 * (torch.exp(2 * l1) + (mq - mp) ** 2) / torch.exp(2 * l2)).sum(1)
        ((z_r * wz).sum() + (kl_r * wk).sum()).backward()
        gpu = [t.detach().to(dev).requires_grad_() for t in cpu]
        z, kl = ops.reparam_kl(gpu[0], gpu[1] * 0.3, gpu[2], gpu[3] * 0.3, eps.to(dev), mode=mode)
        ((z * wz.to(dev)).sum() + (kl * wk.to(dev)).sum()).backward()
        np.testing.assert_allclose(z.detach().cpu().numpy(), z_r.detach().numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(kl.detach().cpu().numpy(), kl_r.detach().numpy(), rtol=2e-5, atol=1e-4)
        for a, b in zip(gpu, cpu):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_linear_and_bmm(dev):
    from pevae_amd import ops
    torch.manual_seed(7)
    for M, K, N in [(256, 512, 64), (6, 16, 8), (100, 768, 300)]:
        x, w, b = torch.randn(M, K), torch.randn(N, K) * 0.1, torch.randn(N)
        g = torch.randn(M, N)
        c = [t.clone().requires_grad_() for t in (x, w, b)]
        (c[0] @ c[1].t() + c[2]).backward(g)
        d = [t.to(dev).requires_grad_() for t in (x, w, b)]
        y = ops.linear(*d)
        y.backward(g.to(dev))
        np.testing.assert_allclose(y.detach().cpu().numpy(), (x @ w.t() + b).numpy(), rtol=1e-5, atol=2e-5)
        for a, r in zip(d, c):
            np.testing.assert_allclose(a.grad.cpu().numpy(), r.grad.numpy(), rtol=1e-4, atol=1e-4)
    a, b2 = torch.randn(3, 5, 16), torch.randn(3, 6, 16)
    ca, cb = a.clone().requires_grad_(), b2.clone().requires_grad_()
    g = torch.randn(3, 5, 6)
    torch.bmm(ca, cb.transpose(1, 2)).backward(g)
    da, db = a.to(dev).requires_grad_(), b2.to(dev).requires_grad_()
    y = ops.bmm(da, db.transpose(1, 2))
    y.backward(g.to(dev))
    np.testing.assert_allclose(y.detach().cpu().numpy(), torch.bmm(a, b2.transpose(1, 2)).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(da.grad.cpu().numpy(), ca.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db.grad.cpu().numpy(), cb.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_adam_clip_guard_matches_torch(dev):
    """pv_grad_guard + pv_adam_step vs clip_grad_norm_ + torch.optim.Adam + constant-with-warmup LambdaLR."""
    from pevae_amd.optim import FlatAdam
    torch.manual_seed(8)
    shapes = [(33, 7), (129,), (5, 5, 3)]
    ps = [torch.randn(*s) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-2)
    warm = 3
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda k: k / max(1.0, warm) if k < warm else 1.0)
    mine = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    fa = FlatAdam(mine, lr=1e-2, max_norm=1.0, warmup_steps=warm, guards=False)
    for step in range(6):
        grads = [torch.randn(*s) * (3.0 if step % 2 else 0.05) for s in shapes]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, 1.0, 2)
        opt.step()
        sched.step()
        fa.zero_grad()
        for p, g in zip(mine, grads):
            p.grad.copy_(g.to(dev))
        fa.step()
        st = fa.state_dict_host()
        assert abs(st["last_norm"] - float(norm)) < 1e-4 * float(norm)
        for a, b in zip(mine, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=2e-6)


def test_adamw_and_grad_scale_match_torch(dev):
    """The freeze-phase optimizer (dialog_models.py:834: torch.optim.AdamW, default weight decay 0.01) and the
    data-parallel 1/world fold: FlatAdam(weight_decay=0.01, grad_scale=1/4) fed 4x the gradient vs AdamW fed the gradient."""
    from pevae_amd.optim import FlatAdam
    torch.manual_seed(12)
    shapes = [(40, 9), (77,)]
    ps = [torch.randn(*s) for s in shapes]
    ref = [p.clone().requires_grad_() for p in ps]
    opt = torch.optim.AdamW(ref, lr=3e-2)
    mine = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    fa = FlatAdam(mine, lr=3e-2, max_norm=2.0, guards=False, weight_decay=0.01, grad_scale=0.25)
    for step in range(5):
        grads = [torch.randn(*s) * (2.0 if step % 2 else 0.1) for s in shapes]
        for p, g in zip(ref, grads):
            p.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, 2.0, 2)
        opt.step()
        fa.zero_grad()
        for p, g in zip(mine, grads):
            p.grad.copy_((4.0 * g).to(dev))
        fa.step()
        assert abs(fa.state_dict_host()["last_norm"] - float(norm)) < 1e-4 * float(norm)
        for a, b in zip(mine, ref):
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=2e-6)


def test_dropout_mask_and_generator(dev):
    """ops.Dropout (encoders.py:54, decoders.py:175,189): with a supplied mask it is exactly x * keep / (1 - p) in both
    passes; with the generator the keep rate is 1 - p, the backward pass regenerates the forward's decisions, a new step
    (pv_rng_advance) or a new call draws a new mask, and the bf16 operand pack sees the same decisions as the fp32 form."""
    from pevae_amd import _lib, ops
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(6, 11, 37, generator=gen)
    keep = torch.rand(6, 11, 37, generator=gen) >= 0.3
    drop = ops.Dropout(0.3).train()
    drop.mask = keep.to(dev)
    xd = x.to(dev).requires_grad_()
    y = drop(xd)
    w = torch.randn(6, 11, 37, generator=gen)
    (y * w.to(dev)).sum().backward()
    np.testing.assert_array_equal(y.detach().cpu().numpy(),
                                  torch.where(keep, x * (1.0 / (1.0 - np.float32(0.3))), torch.zeros(())).numpy())
    np.testing.assert_array_equal(xd.grad.cpu().numpy(),
                                  torch.where(keep, w * (1.0 / (1.0 - np.float32(0.3))), torch.zeros(())).numpy())
    drop.eval()
    assert drop(xd) is xd
    # generator mode
    ops.rng_seed(123, dev)
    drop = ops.Dropout(0.3).train()
    big = torch.ones(64, 50, 512, device=dev, requires_grad=True)
    y1 = drop(big)
    y1.sum().backward()
    kept = y1 != 0
    assert abs(float(kept.float().mean()) - 0.7) < 2e-3
    assert torch.equal(big.grad != 0, kept)                       # backward regenerated the same decisions
    assert float(y1.max()) == pytest.approx(1.0 / 0.7, rel=1e-6)
    y2 = drop(big)                                                # next call of the same module: another site
    assert 0.4 < float(((y2 != 0) == kept).float().mean()) < 0.7  # independent masks agree on ~0.58 of the elements
    site = drop.site_base + 77
    a = ops.DropoutFn.apply(big, 0.3, site, None)
    b = ops.DropoutFn.apply(big, 0.3, site, None)
    assert torch.equal(a, b)                                      # same (seed, step, site): same mask
    ops.rng_advance(dev)
    c = ops.DropoutFn.apply(big, 0.3, site, None)
    assert not torch.equal(a, c)                                  # new step: new mask
    # bf16 pack of a strided [B, L, H] view, time-major rows, against the fp32 form of the same logical tensor
    B, L, H, Bpad, Hp = 5, 7, 40, 32, 64
    src = torch.randn(L, Bpad, H, device=dev)                     # time-major storage, as the recurrence leaves it
    view = src[:, :B].transpose(0, 1)                             # the batch-first module output
    ref = ops.DropoutFn.apply(view, 0.3, site, None)              # logical (b, t, c) indexing
    out = torch.empty(L * Bpad, Hp, dtype=torch.bfloat16, device=dev)
    _lib.call("pv_dropout_pack_bf16", view.data_ptr(), L, B, Bpad, H, view.stride(1), view.stride(0), view.stride(2), H, L * H,
              0.3, None, ops.rng_state(dev).data_ptr(), site, out.data_ptr(), Hp, torch.cuda.current_stream().cuda_stream)
    got = out.view(L, Bpad, Hp)
    assert torch.equal(got[:, :B, :H].transpose(0, 1).float(), ref.to(torch.bfloat16).float())
    assert not got[:, B:].any() and not got[:, :, H:].any()       # padding rows / columns are zero


def test_guard_skips_nonfinite_and_spikes(dev):
    from pevae_amd.optim import FlatAdam
    p = torch.nn.Parameter(torch.ones(64, device=dev))
    fa = FlatAdam([p], lr=0.1, max_norm=5.0, warmup_steps=0, guards=True)
    fa.zero_grad(); p.grad.fill_(0.1); fa.step()
    after1 = p.detach().clone()
    assert not torch.equal(after1, torch.ones(64, device=dev))
    fa.zero_grad(); p.grad.fill_(float("nan")); fa.step()            # main_pretrain.py:213-217
    assert torch.equal(p.detach(), after1) and fa.state_dict_host()["skip"] == 1
    fa.zero_grad(); p.grad.fill_(10.0); fa.step()                    # > 3x running mean, :219-222
    assert torch.equal(p.detach(), after1) and fa.state_dict_host()["skipped_total"] == 2
    fa.zero_grad(); p.grad.fill_(0.1); fa.step()
    assert not torch.equal(p.detach(), after1)


def test_gemm_splitk_exact_and_colsum(dev):
    """Split-K weight-gradient GEMM: exact on integer data (any split count), accumulate form, and the column-sum kernel."""
    import ctypes as C
    from pevae_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    for (M, N, K) in [(96, 40, 1024), (1536, 100, 3200), (130, 512, 640)]:
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        Bm = torch.randint(-2, 3, (N, K), generator=g).float()
        want = A @ Bm.t()
        Ab, Bb = A.to(dev).to(torch.bfloat16), Bm.to(dev).to(torch.bfloat16)
        got = ops.gemm_nt_wgrad(Ab, K, Bb, K, M, N, K)
        assert torch.equal(got.cpu(), want)
        sink = torch.nn.Parameter(torch.zeros(M, N, device=dev))
        sink.grad = torch.ones(M, N, device=dev)
        assert ops.gemm_nt_wgrad(Ab, K, Bb, K, M, N, K, ops.grad_sink(sink, (M, N))) is None
        assert torch.equal(sink.grad.cpu(), want + 1)
        s = C.c_int(0)
        _lib.call("pv_gemm_splitk_plan", M, N, K, 512, C.byref(s))
        assert 1 <= s.value <= K // 64
    x = torch.randn(517, 300, generator=g)
    out = torch.empty(300, device=dev)
    _lib.call("pv_colsum_f32", x.to(dev).data_ptr(), 517, 300, 300, out.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    np.testing.assert_allclose(out.cpu().numpy(), x.sum(0).numpy(), rtol=1e-5, atol=1e-4)


def test_gemm_tn_exact_integers(dev):
    """C = A^T B over K-major operands (the weight-gradient form without transposed copies): exact on integer data with an
    asymmetric pattern in both operands (row / column swaps and fragment-map mistakes show), leading dimensions wider than
    M / N, with and without split-K, returned and accumulated into a gradient sink; equal to the nt form on the copies."""
    import ctypes as C
    from pevae_amd import _lib, ops
    g = torch.Generator().manual_seed(23)
    for (M, N, K, lda, ldb) in [(128, 128, 64, 128, 128), (96, 40, 1024, 96, 64), (1536, 100, 3200, 1536, 128),
                                (136, 512, 640, 192, 512), (1536, 512, 12800, 1536, 512), (8, 3, 64, 8, 8)]:
        A = torch.zeros(K, lda)
        Bm = torch.zeros(K, ldb)
        A[:, :M] = torch.randint(-2, 3, (K, M), generator=g).float() + (torch.arange(M) % 3 == 0).float()
        Bm[:, :N] = torch.randint(-2, 3, (K, N), generator=g).float() + (torch.arange(K).view(-1, 1) % 5 == 0).float()
        if lda > M:
            A[:, M:] = 7.0                       # what lies past M / N inside the leading dimension must not leak into C
        want = A[:, :M].t().double() @ Bm[:, :N].double()
        Ab, Bb = A.to(dev).to(torch.bfloat16), Bm.to(dev).to(torch.bfloat16)
        got = ops.gemm_tn_wgrad(Ab, lda, Bb, ldb, M, N, K)
        assert torch.equal(got.cpu().double(), want), (M, N, K)
        sink = torch.nn.Parameter(torch.zeros(M, N, device=dev))
        sink.grad = torch.ones(M, N, device=dev)
        assert ops.gemm_tn_wgrad(Ab, lda, Bb, ldb, M, N, K, ops.grad_sink(sink, (M, N))) is None
        assert torch.equal(sink.grad.cpu().double(), want + 1), (M, N, K)
        # one split, straight into C (no workspace)
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        _lib.call("pv_gemm_bf16_tn_splitk", Ab.data_ptr(), lda, Bb.data_ptr(), ldb, M, N, K, 1, None, out.data_ptr(), N, 0,
                  torch.cuda.current_stream().cuda_stream)
        assert torch.equal(out.cpu().double(), want), (M, N, K)
    with pytest.raises(_lib.PevaeHipError):
        ops.gemm_tn_wgrad(Ab, 8, Bb, 8, 8, 3, 96)          # K must be a multiple of 64


def test_vocab_nll_skipping_ignored_tokens_matches_dense(dev):
    """Ignored-token compaction (pv_vocab_compact / _gather / live-row bounds / pv_scatter_rows_f32) against the dense
    computation of every row: same per-token NLL, same gradients (ignored rows contribute exactly zero either way)."""
    from pevae_amd import ops
    gen = torch.Generator().manual_seed(13)
    Tn, H, V = 900, 128, 1000
    hid = (torch.randn(Tn, H, generator=gen) * 0.5)
    W = (torch.randn(V, H, generator=gen) * 0.1)
    b = torch.randn(V, generator=gen) * 0.1
    tgt = torch.randint(1, V, (Tn,), generator=gen)
    tgt[torch.rand(Tn, generator=gen) < 0.45] = 0                      # pads scattered through the rows
    tgt[-200:] = 0
    wgt = torch.rand(Tn, generator=gen)
    res = []
    for skip in (True, False):
        ops.SKIP_IGNORED_TOKENS = skip
        try:
            h = hid.to(dev).requires_grad_()
            w = W.to(dev).requires_grad_()
            bb = b.to(dev).requires_grad_()
            nll, _ = ops.VocabNLL.apply(h, w, bb, tgt.to(dev), 1.0 / 0.8, 0, None, None)
            (nll * wgt.to(dev)).sum().backward()
            torch.cuda.synchronize()
        finally:
            ops.SKIP_IGNORED_TOKENS = True
        res.append((nll.detach().cpu(), h.grad.cpu(), w.grad.cpu(), bb.grad.cpu()))
    (n1, h1, w1, b1), (n0, h0, w0, b0) = res
    assert float(n1[tgt == 0].abs().max()) == 0.0 and float(h1[tgt == 0].abs().max()) == 0.0
    np.testing.assert_allclose(n1.numpy(), n0.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(h1.numpy(), h0.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(w1.numpy(), w0.numpy(), rtol=1e-3, atol=1e-4 * float(w0.abs().max()))
    np.testing.assert_allclose(b1.numpy(), b0.numpy(), rtol=1e-3, atol=1e-4 * float(b0.abs().max()))
    # all rows ignored: zero loss, zero gradients, no NaN
    h = hid.to(dev).requires_grad_()
    w = W.to(dev).requires_grad_()
    nll, _ = ops.VocabNLL.apply(h, w, b.to(dev), torch.zeros(Tn, dtype=torch.long, device=dev), 1.0, 0, None, None)
    nll.sum().backward()
    assert float(nll.abs().max()) == 0.0 and float(h.grad.abs().max()) == 0.0 and float(w.grad.abs().max()) == 0.0


def test_gemm_live_rows_and_persistent_walk(dev):
    """The tile walk when there are more tiles than resident workgroups (persistent workgroups, every tile visited exactly
    once), and the device-side row bound with its device-chosen split-K: exact on integer data for several live counts."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(17)
    M, N, K = 4000, 2900, 128                                      # 32 x 23 = 736 tiles > 2 x 256 workgroups
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    Bm = torch.randint(-2, 3, (N, K), generator=g).float()
    got = ops.gemm_nt(A.to(dev).to(torch.bfloat16), K, Bm.to(dev).to(torch.bfloat16), K, M, N, K)
    assert torch.equal(got.cpu(), A @ Bm.t())
    M, N, K = 3000, 200, 2048
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    Bm = torch.randint(-2, 3, (N, K), generator=g).float()
    want = A @ Bm.t()
    Ab, Bb = A.to(dev).to(torch.bfloat16), Bm.to(dev).to(torch.bfloat16)
    for live in (0, 1, 128, 129, 1500, 3000, 5000):
        cnt = torch.tensor([live], dtype=torch.int32, device=dev)
        out = ops.gemm_nt_live_rows(Ab, K, Bb, K, M, N, K, cnt)
        n = min(live, M)
        assert torch.equal(out[:n].cpu(), want[:n]), live


def test_dp_sparse_pack_merge_on_device(dev):
    """dp.pack_sparse_rows / merge_sparse_rows (the user / item table exchange of the data-parallel step) on the device:
    two simulated ranks' packed contributions merged through pv_embedding_bwd equal the dense sum of their tables."""
    from pevae_amd import dp
    g = torch.Generator().manual_seed(21)
    rows, D, n = 300, 256, 64
    tables, ids = [], []
    for r in range(2):
        idv = torch.randint(0, rows, (n,), generator=g)
        idv[5] = idv[9] = idv[0]
        t = torch.zeros(rows, D)
        t[idv.unique()] = torch.randn(idv.unique().numel(), D, generator=g)
        tables.append(t)
        ids.append(idv)
    packed = [dp.pack_sparse_rows(t.to(dev), i.to(dev)) for t, i in zip(tables, ids)]
    ids_all = torch.cat([p[0] for p in packed])
    rows_all = torch.cat([p[1] for p in packed])
    for r in range(2):
        mine = tables[r].to(dev).clone()
        dp.merge_sparse_rows(mine, ids[r].to(dev), ids_all, rows_all)
        torch.testing.assert_close(mine.cpu(), tables[0] + tables[1], rtol=1e-6, atol=1e-6)
        if r == 0:
            first = mine.clone()
    assert torch.equal(first, mine)                        # both "ranks" end with the same bits


def test_dp_sparse_pack_merge_kernels(dev):
    """pv_sparse_pack / pv_sparse_merge (the exchange's device path: one pack launch, one all-gather, one merge launch per
    table): three simulated ranks' blocks laid out as the all-gather leaves them; every rank's merged table equals the
    sum of the three tables (sequential fp32 adds in rank order: bit-exact against the same adds in torch) and the pack
    equals dp.pack_sparse_rows."""
    from pevae_amd import _lib, dp, ops
    g = torch.Generator().manual_seed(22)
    rows, D, n, W = 500, 256, 64, 3
    _sparse_pack_merge_case(dev, g, rows, D, n, W)
    _sparse_pack_merge_case(dev, g, 30000, 256, 256, 8)      # the N = 8 Yelp shape: 2048 entries


def _sparse_pack_merge_case(dev, g, rows, D, n, W):
    from pevae_amd import _lib, dp, ops
    tables, ids = [], []
    for r in range(W):
        idv = torch.randint(0, rows, (n,), generator=g)
        idv[5] = idv[9] = idv[0]
        idv[20] = 7                                            # id 7 on every rank
        t = torch.zeros(rows, D)
        t[idv.unique()] = torch.randn(idv.unique().numel(), D, generator=g)
        tables.append(t.to(dev))
        ids.append(idv.to(dev))
    S = n * (D + 1) + 40                                       # a second table's block would follow: stride > block
    recv = torch.zeros(W * S, device=dev)
    for r in range(W):
        blk = recv[r * S:(r + 1) * S]
        _lib.call("pv_sparse_pack", ops._p(tables[r]), rows, D, ops._p(ids[r]), n, ops._p(blk[:n]), ops._p(blk[n:n + n * D]),
                  ops._stream())
        pid, prow = dp.pack_sparse_rows(tables[r], ids[r])
        assert torch.equal(blk[:n], pid.float()) and torch.equal(blk[n:n + n * D].view(n, D), prow)
    want = tables[0].clone()
    for r in range(1, W):
        want = want + tables[r]
    outs = []
    for r in range(W):
        mine = tables[r].clone()
        _lib.call("pv_sparse_merge", ops._p(recv), W, S, n, D, rows, ops._p(mine), ops._stream())
        assert torch.equal(mine, want), r
        outs.append(mine)
    with pytest.raises(RuntimeError):
        _lib.call("pv_sparse_pack", ops._p(tables[0]), (1 << 24) + 1, D, ops._p(ids[0]), n, ops._p(recv), ops._p(recv), ops._stream())


@pytest.mark.parametrize("R,C,lds", [(200, 96, 96), (130, 512, 520), (77, 100, 100), (64, 36, 40), (300, 30, 30)])
def test_transpose_to_bf16_both_paths(dev, R, C, lds):
    """pv_transpose_to_bf16 from fp32 and from bf16 sources: shapes that take the 16-byte path and shapes that fall back
    to the element-wise kernel, zero fill of the padded columns; bit-exact (a cast and a copy)."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(R + C)
    src = torch.randn(R, lds, generator=g)
    rpad = ops.round_up(R, 64)
    for dt in (torch.float32, torch.bfloat16):
        s = src.to(dt).to(dev)
        out = torch.full((C, rpad), 7.0, dtype=torch.bfloat16, device=dev)
        ops.transpose_to_bf16(s, R, C, lds, rpad, out=out)
        want = torch.zeros(C, rpad, dtype=torch.bfloat16)
        want[:, :R] = src[:, :C].to(dt).to(torch.bfloat16).t()
        assert torch.equal(out.cpu(), want), (dt, R, C)


def test_gemm_transposed_store_exact(dev):
    """pv_gemm_bf16_nt_ct: C^T + bias per column of C, exact on integer data, ragged N and a leading dimension > M."""
    from pevae_amd import ops
    g = torch.Generator().manual_seed(23)
    for (M, N, K, ld) in [(256, 100, 64, 256), (1000, 1536, 128, 1024), (132, 70, 192, 200)]:
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        Bm = torch.randint(-2, 3, (N, K), generator=g).float()
        bias = torch.randint(-3, 4, (N,), generator=g).float()
        out = torch.full((N, ld), 9.0, device=dev)
        ops.gemm_nt_ct(A.to(dev).to(torch.bfloat16), K, Bm.to(dev).to(torch.bfloat16), K, M, N, K, bias.to(dev), out)
        assert torch.equal(out[:, :M].cpu(), (A @ Bm.t() + bias).t())
        assert float((out[:, M:] - 9.0).abs().max()) == 0.0 if ld > M else True


def test_nll_head_matches_the_scalar_ops(dev):
    """ops.NllHead: {total / batch, exp(total / count)} and its gradient against the stock scalar arithmetic
    (criterions.py:64 and :137)."""
    from pevae_amd import ops
    sums = torch.tensor([5231.75, 7050.0], device=dev, requires_grad=True)
    ref = sums.detach().clone().requires_grad_()
    nll, ppl = ops.NllHead.apply(sums, 256)
    nll_r, ppl_r = ref[0] / 256, torch.exp(ref[0] / ref[1])
    np.testing.assert_allclose(float(nll), float(nll_r), rtol=1e-6)
    np.testing.assert_allclose(float(ppl), float(ppl_r), rtol=1e-5)
    (3.0 * nll + 0.5 * ppl).backward()
    (3.0 * nll_r + 0.5 * ppl_r).backward()
    np.testing.assert_allclose(float(sums.grad[0]), float(ref.grad[0]), rtol=1e-5)
    assert float(sums.grad[1]) == 0.0          # the token count carries no gradient (a constant of the batch)
    sums.grad = None
    nll2, _ = ops.NllHead.apply(sums, 256)
    nll2.backward()                            # only one of the two outputs used
    np.testing.assert_allclose(float(sums.grad[0]), 1.0 / 256, rtol=1e-6)


def test_fan_out_sums_the_gradients_of_its_aliases(dev):
    """ops.FanOut / pv_sum_group: aliases cost no forward launch, the backward adds what arrived per tensor left to
    right (bit-exact against the same fp32 adds), unused aliases and single-use tensors pass through."""
    from pevae_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    xs = [torch.randn(sh, generator=g).to(dev).requires_grad_() for sh in ((256, 512), (256, 128), (10, 64), (7, 12))]
    ws = [[torch.randn(x.shape, generator=g).to(dev) for _ in range(4)] for x in xs]
    al = ops.FanOut.apply(4, *xs)
    assert all(a.data_ptr() == xs[i // 4].data_ptr() for i, a in enumerate(al))
    uses = (4, 3, 2, 1)                                      # the last tensor has one reader: no sum
    loss = sum((al[4 * i + j] * ws[i][j]).sum() for i in range(4) for j in range(uses[i]))
    loss.backward()
    for i, x in enumerate(xs):
        want = ws[i][0].clone()
        for j in range(1, uses[i]):
            want = want + ws[i][j]
        assert torch.equal(x.grad, want), i
    # the dispenser: unknown tensors and exhausted alias lists fall back to the tensor itself
    y = torch.randn(8, 16, device=dev, requires_grad=True)
    fan = ops.Fan(2, [y, y, None])
    a, b, c = fan(y), fan(y), fan(y)
    assert a is not y and b is not y and c is y and fan(xs[0]) is xs[0]
    (a.sum() + 2 * b.sum() + 3 * c.sum()).backward()
    assert torch.equal(y.grad, torch.full_like(y, 6.0))


@pytest.mark.parametrize("use_mask", [True, False])
def test_scatter_rows_scaled_with_dropout_backward_folded_in(dev, use_mask):
    """pv_scatter_rows_scaled_drop_f32 = pv_scatter_rows_scaled_f32 followed by pv_dropout_f32 on the destination (the
    decoder's output dropout backward, decoders.py:189), bit for bit: supplied keep mask and regenerated Philox decisions,
    time-major rows with batch padding (n1p > n1), a live-row count below T."""
    from pevae_amd import _lib, ops
    g = torch.Generator().manual_seed(31)
    L, B, Bpad, H, p = 7, 24, 32, 64, 0.3
    T = L * Bpad
    live = 150
    src = torch.randn(T, H, generator=g).to(dev)
    scale = torch.randn(T, generator=g).to(dev)
    perm = torch.randperm(T, generator=g).to(dev)
    count = torch.tensor([live], dtype=torch.int32, device=dev)
    mask = (torch.rand(B, L, H, generator=g) >= p).to(torch.uint8).to(dev) if use_mask else None
    m0, m1 = H, L * H                                   # operand row (t, b) holds logical [b, t, :] of a [B, L, H] tensor
    ops.rng_seed(1234, dev)
    want = torch.empty(T, H, device=dev)
    _lib.call("pv_scatter_rows_scaled_f32", ops._p(src), ops._p(scale), 0.5, ops._p(perm), ops._p(count), T, H, ops._p(want), H,
              ops._stream())
    _lib.call("pv_dropout_f32", ops._p(want), L, B, H, Bpad * H, H, 1, m0, m1, p, ops._p(mask), ops._p(ops.rng_state(dev)), 77,
              ops._p(want), Bpad * H, H, ops._stream())
    got = torch.full((T, H), 9.0, device=dev)
    _lib.call("pv_scatter_rows_scaled_drop_f32", ops._p(src), ops._p(scale), 0.5, ops._p(perm), ops._p(count), T, H, ops._p(got), H,
              B, Bpad, m0, m1, p, ops._p(mask), ops._p(ops.rng_state(dev)), 77, ops._stream())
    v = got.view(L, Bpad, H)
    assert torch.equal(v[:, :B], want.view(L, Bpad, H)[:, :B])
    assert float(v[:, B:].abs().max()) == 0.0            # padding rows: zeros
    kept = float((v[:, :B] != 0).float().mean())
    assert 0.05 < kept < 0.7                             # ~ live / T * (1 - p) of the entries survive
