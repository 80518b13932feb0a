"""GPU parity of the DLS latent pyramid (a15-a17): Block / DlsRec / DlsPri against the goldens generated from the
reference's own classes, the DLS_Auto helpers against dls_helpers.npz, and DLSAuto + the composed 'dls' step against
the CPU oracle.  Everything here is fp32 (f32-input MFMA for the linears): tolerance 1e-4 relative."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, load_params
from oracle import ref_modules as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def default_block_kernels():
    """Every test starts and ends with the default fused Block kernels (a test may select the 16-row ones)."""
    yield
    from pevae_amd import ops
    ops.block_rows(4, 1)


def T(a, dev):
    return torch.from_numpy(np.array(a)).to(dev)


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_block_golden(dev):
    from pevae_amd import dls_latent as L
    g = load_golden("block")
    blk = load_params(L.Block(32, 16, 24), g).to(dev)
    blk.train()
    x = T(g["x"], dev).requires_grad_()
    y = blk(x)
    (y * T(g["w"], dev)).sum().backward()
    close(y, g["y"])
    close(x.grad, g["dx"])
    for k, p in blk.named_parameters():
        close(p.grad, g["g." + k], rtol=3e-4, atol=2e-5)
    for k, v in blk.state_dict().items():           # power-iteration buffers advanced identically
        close(v, g["post." + k], rtol=1e-4, atol=1e-6)
    close(blk(T(g["x"], dev)), g["y2"])
    blk.eval()
    close(blk(T(g["x"], dev)), g["y_eval"])


@pytest.mark.parametrize("M,din,dout,mid,sinks", [(37, 48, 32, 96, False), (256, 128, 64, 64, True), (16, 16, 16, 16, False),
                                                  (70, 512, 256, 256, True), (256, 256, 512, 512, False),
                                                  (250, 192, 256, 256, True), (3, 32, 16, 16, True), (129, 96, 128, 128, False)])
@pytest.mark.parametrize("rows", [4, 16])
def test_block_fused_matches_unfused(dev, M, din, dout, mid, sinks, rows):
    """The one-launch Block kernels (pv_block_fwd / pv_block_bwd; rows = 4: the 4-row slabs that are the default, 16: the
    round-2 16-row slabs and clusters) against the per-linear path that the reference goldens pin (test_block_golden), on
    ragged row counts, with gradients returned and accumulated into existing .grad."""
    import copy
    from pevae_amd import dls_latent as L, ops
    torch.manual_seed(3)
    ops.block_rows(rows)
    blk_a = L.Block(din, dout, mid).to(dev).train()
    blk_b = copy.deepcopy(blk_a)
    assert ops.block_fusable(din, mid, dout)
    x = torch.randn(M, din, device=dev)
    w = torch.randn(M, dout, device=dev)
    res = []
    for blk, fused in ((blk_a, True), (blk_b, False)):
        ops.FUSED_BLOCK = fused
        try:
            if sinks:
                for p in blk.parameters():
                    p.grad = torch.full_like(p, 0.25)
            xi = x.clone().requires_grad_()
            y = blk(xi)
            (y * w).sum().backward()
        finally:
            ops.FUSED_BLOCK = True
        res.append((y, xi.grad, blk))
    (ya, dxa, _), (yb, dxb, _) = res
    close(ya, yb, rtol=1e-4, atol=2e-5)
    close(dxa, dxb, rtol=1e-4, atol=1e-4 * float(dxb.abs().max()))
    for (k, p), (_, q) in zip(blk_a.named_parameters(), blk_b.named_parameters()):
        close(p.grad, q.grad, rtol=1e-4, atol=1e-4 * float(q.grad.abs().max()) + 1e-6)
    for (k, u), (_, v) in zip(blk_a.named_buffers(), blk_b.named_buffers()):
        close(u, v, rtol=1e-5, atol=1e-6)
    ops.block_check()          # no hand-off between the workgroups of a slab (512 / 256-wide Blocks) gave up


@pytest.mark.parametrize("name", ["dlsrec", "dlspri"])
def test_pyramid_golden(dev, name):
    from pevae_amd import dls_latent as L
    g = load_golden(name)
    m = load_params((L.DlsRec if name == "dlsrec" else L.DlsPri)(64, 16), g).to(dev)
    m.eval()
    out, acts = m(T(g["x"], dev))
    assert len(acts) == int(g["n_acts"])
    close(out, g["out"])
    for i, a in enumerate(acts):
        close(a, g["act%d" % i])


def test_cons_loss_and_kl_golden(dev):
    from pevae_amd import ops
    g = load_golden("dls_helpers")
    v1, v2 = T(g["v1"], dev).requires_grad_(), T(g["v2"], dev).requires_grad_()
    cl = ops.cons_loss(v1, v2)
    cl.backward()
    close(cl, g["cons"], rtol=1e-5)
    close(v1.grad, g["dv1"], rtol=1e-4, atol=1e-6)
    close(v2.grad, g["dv2"], rtol=1e-4, atol=1e-6)
    ts = [T(g[k], dev).requires_grad_() for k in ("m1", "m2", "l1", "l2")]
    _, kl = ops.reparam_kl(ts[0], ts[2] * 0.3, ts[1], ts[3] * 0.3, None, mode=1, want_z=False)
    ops.mean(kl).backward()
    close(ops.mean(kl), g["kl"], rtol=1e-5)
    for n, t in zip(("dm1", "dm2", "dl1", "dl2"), ts):
        close(t.grad, g[n], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("restarted,training", [(False, True), (True, True), (True, False)])
def test_dls_auto_matches_oracle(dev, restarted, training):
    from argparse import Namespace
    from pevae_amd import dls_latent as L
    cfg = Namespace(expand_weight=0.5, converge_weight=0.1, cip_weight=1.0, ost=False)
    torch.manual_seed(3)
    ref = R.DLSAuto(4, 16, cfg)
    gpu = L.DLSAuto(4, 16, cfg)
    gpu.load_state_dict(ref.state_dict())
    gpu = gpu.to(dev)
    ref.train(training)
    gpu.train(training)
    B = 12
    gen = torch.Generator().manual_seed(4)
    ins = [torch.randn(B, 4, generator=gen), torch.randn(B, 4, generator=gen)]
    act1 = [torch.randn(B, d, generator=gen) for d in (32, 16, 8)]       # [-1] is the deepest level
    act2 = [torch.randn(B, d, generator=gen) for d in (32, 16, 8)]
    noise = [torch.rand(B, d, generator=gen) for d in (8, 16, 32)]
    c = [t.clone().requires_grad_() for t in ins + act1 + act2]
    z_r, rkl_r = ref(c[0], c[1], c[2:5], c[5:8], restarted, noise=noise)
    (z_r.sum() * 0.1 + rkl_r).backward()
    d = [t.to(dev).requires_grad_() for t in ins + act1 + act2]
    z, rkl = gpu(d[0], d[1], d[2:5], d[5:8], restarted, noise=[n.to(dev) for n in noise])
    (z.sum() * 0.1 + rkl).backward()
    close(z, z_r.detach(), rtol=2e-4, atol=2e-5)
    close(rkl, rkl_r.detach(), rtol=2e-4, atol=2e-5)
    # gradients: 5e-4 relative plus 1e-4 of the tensor's largest entry (fp32 sums in a different order)
    for a, b in zip(d, c):
        if b.grad is not None:
            close(a.grad, b.grad, rtol=5e-4, atol=5e-5 + 1e-4 * float(b.grad.abs().max()))
    ref_p = dict(ref.named_parameters())
    for n, p in gpu.named_parameters():
        if ref_p[n].grad is not None:
            close(p.grad, ref_p[n].grad, rtol=1e-3, atol=5e-5 + 1e-4 * float(ref_p[n].grad.abs().max()))


def test_den_auto_golden(dev):
    """DeNAuto (DEM discriminator + Self-Adaption Fusion, row a17) against the golden made from the reference's own
    DeN_Auto class: z, rkl, every input / parameter gradient, the advanced power-iteration buffers."""
    import types
    from pevae_amd import dls_latent as L
    g = load_golden("den_auto")
    cfg = types.SimpleNamespace(expand_weight=0.5, converge_weight=0.1, ost=False, no_mutual_information=False)
    m = load_params(L.DeNAuto(16, 32, cfg), g).to(dev)
    m.train()
    output = T(g["output"], dev).requires_grad_()
    act1 = [T(g["act1_%d" % i], dev).requires_grad_() for i in range(2)]
    act2 = [T(g["act2_%d" % i], dev).requires_grad_() for i in range(2)]
    z, rkl = m(output, act1, act2, False, [T(g["noise_%d" % i], dev) for i in range(2)],
               [T(g["gumbel_%d" % i], dev) for i in range(2)])
    (z.sum() * 0.01 + rkl).backward()
    close(z, g["z"], rtol=1e-4, atol=1e-5)
    close(rkl, g["rkl"], rtol=1e-4, atol=1e-5)
    close(output.grad, g["d_output"], rtol=1e-4, atol=1e-4 * float(np.abs(g["d_output"]).max()))
    for i in range(2):
        close(act1[i].grad, g["d_act1_%d" % i], rtol=1e-4, atol=1e-4 * float(np.abs(g["d_act1_%d" % i]).max()))
        close(act2[i].grad, g["d_act2_%d" % i], rtol=1e-4, atol=1e-4 * float(np.abs(g["d_act2_%d" % i]).max()))
    for k, p in m.named_parameters():
        close(p.grad, g["g." + k], rtol=3e-4, atol=1e-4 * float(np.abs(g["g." + k]).max()) + 1e-6)
    for k, v in m.state_dict().items():
        if "post." + k in g:
            close(v, g["post." + k], rtol=1e-4, atol=1e-6)
    m.eval()                                     # inference branch (:2761): prior means only, no random draw
    z2, rkl2 = m(T(g["output"], dev), [T(g["act1_%d" % i], dev) for i in range(2)],
                 [T(g["act2_%d" % i], dev) for i in range(2)], True)
    ref = load_params(R.DeNAuto(16, 32, cfg), {k: v for k, v in g.items()})
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    ref.eval()
    z3, _ = ref(T(g["output"], "cpu"), [T(g["act1_%d" % i], "cpu") for i in range(2)],
                [T(g["act2_%d" % i], "cpu") for i in range(2)], True)
    close(z2, z3, rtol=1e-4, atol=1e-5)


def test_cons_loss_sum_matches_single_calls(dev):
    """The batched cons_loss (pv_cons_loss_fwd / _bwd: all pairs of a step in one launch per stage) against the sum of
    the single-pair path (which is golden-tested above): value and every input gradient."""
    from pevae_amd import ops
    gen = torch.Generator().manual_seed(31)
    B = 48
    dims = [8, 64, 32, 16, 12]
    pairs_a = [(torch.randn(B, d, generator=gen).to(dev).requires_grad_(), torch.randn(B, d, generator=gen).to(dev).requires_grad_())
               for d in dims]
    pairs_b = [(a.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()) for a, b in pairs_a]
    tot_a = ops.cons_loss_sum(pairs_a)
    tot_b = sum(ops.cons_loss(a, b) for a, b in pairs_b)
    (tot_a * 0.7).backward()
    (tot_b * 0.7).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(tot_a.detach()), float(tot_b.detach()), rtol=1e-5, atol=1e-5)
    for (a, b), (c, d) in zip(pairs_a, pairs_b):
        np.testing.assert_allclose(a.grad.cpu().numpy(), c.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(b.grad.cpu().numpy(), d.grad.cpu().numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("tag", ["train", "restarted_train", "restarted_eval", "train_gir"])
def test_dls_auto_golden_from_reference_class(dev, tag):
    """L.DLSAuto (HIP) against the golden made from the reference's own DLS_Auto class (dialog_models.py:3274-3352):
    z, rkl, every input / parameter gradient, the advanced power-iteration vectors.  train_gir: with the density-estimate
    regulariser (:3232-3236, :3311) switched on in the reference; the HIP path adds its closed form."""
    from pevae_amd import dls_latent as L
    from test_oracle_golden import dls_auto_from_golden, run_dls_auto_golden
    g = load_golden("dls_auto_" + tag)
    run_dls_auto_golden(dls_auto_from_golden(L.DLSAuto, g, dev), g, lambda a: T(a, dev), 5e-4, 1e-4)


def test_global_negatives_two_ranks(dev, tmp_path):
    """config.global_negatives (dialog_models.py:3266-3272, :2676-2685, :2722-2726): two ranks of 8 rows each
    -- DLSAuto's cons_loss over gathered rows, DeNAuto's hard negatives drawn among all 16 rows -- give the loss and the
    rank-averaged parameter gradient of ONE process at 16 rows.  Both ranks use this box's one GPU over a gloo group
    (dp.GatherRows stages through the host there); tests/test_dp_gloo.py covers the gradient rule on CPU."""
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _global_negatives_worker as W
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_global_negatives_worker.py")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(port), str(tmp_path)]) for r in range(2)]
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=100))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(-9)
    assert codes == [0, 0]
    dls, den = W.build(dev, False)                     # one process, 16 rows, the reference's per-batch form
    d = W.inputs(16, dev)
    d["_gumbels"] = [g.to(dev) for g in d["gumbels"]]
    want_loss, want_flat = W.run(dls, den, d, 0, 16, dev)
    scale = float(want_flat.abs().max())
    for r in range(2):
        loss, flat = torch.load(os.path.join(str(tmp_path), "gn%d.pt" % r), weights_only=True)
        np.testing.assert_allclose(float(loss), float(want_loss), rtol=2e-5)
        np.testing.assert_allclose(flat.numpy(), want_flat.numpy(), rtol=1e-3, atol=2e-5 * scale)
    # the per-shard default on the same halves is a different loss
    dls, den = W.build(dev, False)
    d["_gumbels"] = [g[:8, :8].to(dev) for g in d["gumbels"]]
    half, _ = W.run(dls, den, d, 0, 8, dev)
    assert abs(float(half) - float(want_loss)) > 1e-3
