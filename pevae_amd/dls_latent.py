"""The DLS latent stack of the reference on the HIP kernels: Block (dialog_models.py:253-271), DlsRec / DlsPri
(:1835-1870) and DLS_Auto (:3161-3352), with the reference's parameter names (`c1.weight_orig`, `c1.weight_u`,
`layers.0.0.c1.bias`, ...) so its checkpoints load.

Deviations, all forced by the reference being un-runnable as written:
  * DLS_Auto's density-estimate regulariser (:3232-3236, :3308, :3311) takes torch.log of a Python int
    (`n_batch = -1`) and raises, and config.gir_weight is defined by no parser; it is computed when config.gir_weight is
    non-zero and `n_batch` has been set to a positive number, in closed form (see DLSAuto.density_term).
  * `gau_list` / `batch_arrange` need CUDA and a fixed batch size at construction (:3186, :3198); the InfoNCE targets
    are arange(actual batch) here.
  * uniform noise for `dis_reparameterization` (:3206-3207) can be passed in (`noise=[...]`) so that parity tests are
    deterministic; by default it is drawn on the device.
"""
import os

import torch
import torch.nn as nn

from . import dp, ops

# DLS_Auto's batched consistency losses on a third stream beside the first level (forward) and at the head of the backward
CONS_BESIDE = os.environ.get("PEVAE_CONS_BESIDE", "1") == "1"


class SNLinear(nn.Module):
    """spectral_norm(nn.Linear(in, out)) with torch.nn.utils.spectral_norm's state_dict layout."""

    def __init__(self, in_f, out_f, eps=1e-12):
        super().__init__()
        lin = nn.Linear(in_f, out_f)
        self.weight_orig = nn.Parameter(lin.weight.data.clone())
        self.bias = nn.Parameter(lin.bias.data.clone())
        self.eps = eps
        self.register_buffer("weight_u", nn.functional.normalize(torch.randn(out_f), dim=0, eps=eps))
        self.register_buffer("weight_v", nn.functional.normalize(torch.randn(in_f), dim=0, eps=eps))

        self._inv_sigma_pre = None      # set by ops.spectral_prepass for exactly one forward
        self.clone_uv = False           # True for modules that run more than once per step (DeNAuto's discriminator)

    def forward(self, x, x_act=None, emit_gelu=False):
        """y = (x_act if given else x) @ (W / sigma)^T + b.  x_act = gelu(x) written by the producer of x; the gradient
        flows to x through gelu'.  emit_gelu: return (y, gelu(y)) so that the next linear gets its activation for free."""
        pre, self._inv_sigma_pre = self._inv_sigma_pre, None
        y, gy = ops.sn_linear(x, self.weight_orig, self.bias, self.weight_u, self.weight_v, self.training, self.eps,
                              x_act=x_act, emit_gelu=emit_gelu, inv_sigma=pre, clone_uv=self.clone_uv)
        return (y, gy) if emit_gelu else y


class Block(nn.Module):
    def __init__(self, input_dim, output_dim, mid_dim=None):
        super().__init__()
        mid = input_dim // 2 if mid_dim is None else mid_dim
        self.c1 = SNLinear(input_dim, input_dim)
        self.c2 = SNLinear(input_dim, mid)
        self.c3 = SNLinear(mid, mid)
        self.c4 = SNLinear(mid, input_dim)
        self.layer_norm = nn.LayerNorm(input_dim)
        self.translate = SNLinear(input_dim, output_dim)

    def forward(self, x, x_act=None, emit_gelu=False):
        """x_act: gelu(x) if the producer of x already wrote it (else one standalone GELU launch); emit_gelu: also
        return gelu(output) for a following Block.  Each GELU is computed once, in the epilogue of the linear that
        produces its argument."""
        if ops.block_fusable(self.c1.weight_orig.shape[1], self.c2.weight_orig.shape[0], self.translate.weight_orig.shape[0]):
            out = ops.block_forward(self, x)                 # one launch; gelu(x) is formed inside the kernel
            return (out, None) if emit_gelu else out
        if x_act is None:
            x_act = ops.gelu(x).detach()          # data for the GEMM; the gradient reaches x through c1's gelu' epilogue
        y, g = self.c1(x, x_act, emit_gelu=True)
        y, g = self.c2(y, g, emit_gelu=True)
        y, g = self.c3(y, g, emit_gelu=True)
        y = self.c4(y, g)
        out = ops.add_layer_norm(x, y, self.layer_norm.weight, self.layer_norm.bias, self.layer_norm.eps)
        return self.translate(out, emit_gelu=emit_gelu)


def block_dims(blk):
    return blk.c1.weight_orig.shape[1], blk.c2.weight_orig.shape[0], blk.translate.weight_orig.shape[0]


def run_blocks(blks, xs):
    """Independent Blocks: one fused launch per group when every one of them has a fused kernel, else parallel stream
    branches of the per-Block path."""
    if all(ops.block_fusable(*block_dims(b)) for b in blks):
        return ops.blocks_forward(list(blks), list(xs))
    return ops.parallel(*[(lambda b=b, x=x: b(x)) for b, x in zip(blks, xs)])


def pyramid_pair(pyr_a, x_a, pyr_b, x_b):
    """Two pyramids of equal depth walked level by level, the two Blocks of a level sharing a launch
    (DlsPri / DlsRec on prior_h / h, dialog_models.py:3529-3530).  -> ((out_a, acts_a), (out_b, acts_b))"""
    la = [l[0] if isinstance(l, nn.Sequential) else l for l in pyr_a.layers]
    lb = [l[0] if isinstance(l, nn.Sequential) else l for l in pyr_b.layers]
    if len(la) != len(lb):
        return ops.parallel(lambda: pyr_a(x_a), lambda: pyr_b(x_b))
    if x_a.dim() == 3:
        x_a = x_a[:, 0]
    if x_b.dim() == 3:
        x_b = x_b[:, 0]
    acts_a, acts_b = [], []
    for ba, bb in zip(la, lb):
        # an activation feeds the next Block and whoever reads the act lists: one alias each, so that the two
        # gradients of both pyramids meet in one launch of the backward pass (ops.FanOut)
        fan = ops.Fan(2, [x_a, x_b])
        acts_a.append(fan(x_a))
        acts_b.append(fan(x_b))
        x_a, x_b = run_blocks([ba, bb], [fan(x_a), fan(x_b)])
    return (x_a, acts_a), (x_b, acts_b)


class _Pyramid(nn.Module):
    def __init__(self, input_dim, output_dim, wrap):
        super().__init__()
        self.layers = nn.ModuleList()
        d = input_dim
        while d >= output_dim:
            blk = Block(d, d // 2, d // 2)
            self.layers.append(nn.Sequential(blk) if wrap else blk)
            d //= 2

    def forward(self, output):
        if output.dim() == 3:
            output = output[:, 0]            # the reference takes the [CLS] row (:1846, :1864)
        acts, g = [], None
        for layer in self.layers:
            acts.append(output)
            blk = layer[0] if isinstance(layer, nn.Sequential) else layer
            output, g = blk(output, g, emit_gelu=True)       # each Block hands gelu(output) to the next one
        return output, acts


class DlsRec(_Pyramid):
    def __init__(self, input_dim, output_dim, config=None):
        super().__init__(input_dim, output_dim, wrap=False)


class DlsPri(_Pyramid):
    def __init__(self, input_dim, output_dim, config=None):
        super().__init__(input_dim, output_dim, wrap=True)


def spectral_modules(root):
    return [m for m in root.modules() if isinstance(m, SNLinear)]


class DLSAuto(nn.Module):
    """DLS_Auto.forward(output, rec_output, act1, act2, restarted) -> (z, rkl)  (dialog_models.py:3274-3352)."""

    def __init__(self, input_dim, output_dim, config):
        super().__init__()
        self.config = config
        self.n_batch = -1                                  # :3169; the density term needs a positive number here
        self.layers, self.dis_layers, self.dis_layers2 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        cnt = 0
        while input_dim <= output_dim:
            self.layers.append(Block(input_dim, input_dim * 2, input_dim * 2))
            k = 2 if cnt == 0 else 3
            self.dis_layers.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            self.dis_layers2.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            input_dim *= 2
            cnt += 1

    def density_term(self, batch):
        """|qgroup - density_estimate(concatenated levels)| * gir_weight (dialog_models.py:3308, :3311).
        density_estimate (:3232-3236) is  mean_a[ sum_b sum_d log N((s[a,d] - m[b,d]) / v[b,d]; 0, 1) ] - log(n_batch * B):
        the double sum is additive over the feature axis, so the sums of the per-level calls (qgroup) cancel against the
        one call on the concatenated features and what is left is the constant (levels - 1) * log(n_batch * B) -- no
        kernel, no gradient (the reference's own value differs from it by fp32 rounding of the two ~1e3 sums: 3e-5 at
        B = 10; pinned by tests/golden/dls_auto_train_gir.npz)."""
        gir = float(getattr(self.config, "gir_weight", 0.0) or 0.0)
        if gir == 0.0:
            return 0.0
        n_batch = float(self.n_batch)
        if not n_batch > 0:
            raise ValueError("DLS_Auto: gir_weight != 0 needs n_batch > 0 (the reference's n_batch = -1, "
                             "dialog_models.py:3169, makes its torch.log raise)")
        import math
        return abs(gir * (len(self.layers) - 1) * math.log(n_batch * batch))

    def forward(self, output, rec_output, act1, act2, restarted=False, noise=None):
        cfg = self.config
        cip, ew, cw = getattr(cfg, "cip_weight", 1.0), cfg.expand_weight, cfg.converge_weight
        scaled = (lambda v: v) if cip == 1.0 else (lambda v: v * cip)     # cip_weight = 1 (the default): no launch
        batched = not restarted and not getattr(cfg, "ost", False) and all(
            a.shape[1] % 4 == 0 for a in list(act1) + [output]) and len(self.layers) < 7
        use, cons_beside, held = (lambda t: t), None, None
        # config.global_negatives under batch sharding: the score matrices of cons_loss span the GLOBAL batch (every rank
        # computes the same loss from gathered rows; dp.GatherRows), as one process at that batch would
        gather, _ = dp.global_rows(cfg)
        glob = (lambda t: t) if gather is None else (lambda t: gather(t, replicated=True))
        if batched:
            # every cons_loss of the train branch (:3284 and :3307 per level) reads only the pyramids' activations: one
            # batched call up front instead of one chain of ten launches per level
            nl = len(self.layers)
            # `output` and the activations of the fused levels have 2-4 readers in here (cons_loss, the level's Blocks,
            # two level kernels): one alias per reader, their gradients summed by one launch (ops.FanOut)
            use = ops.Fan(4, [output] + [a[-i - 1] for i in range(nl) for a in (act1, act2)])
            pairs = ([(glob(use(output)), glob(rec_output))] +
                     [(glob(use(act1[-i - 1])), glob(use(act2[-i - 1]))) for i in range(nl)])
            if CONS_BESIDE and gather is None and output.is_cuda and ops.PRESORT_OK:   # (one capture for both passes)
                # ... and nothing in the levels reads it: on a third stream beside the first level (and its backward at
                # the head of the backward pass instead of between the levels' and the pyramids'); it meets the
                # running loss behind the first level kernel, in the order it had: (cons + kl_0) + kl_1 + ...
                with ops.fork_third():
                    cons_beside = scaled(ops.cons_loss_sum(pairs))
                rkl = None
            else:
                rkl = scaled(ops.cons_loss_sum(pairs))
        else:
            rkl = scaled(ops.cons_loss(glob(output), glob(rec_output)))
        last_sample = last_t = nxt = None
        if noise is None and not restarted and not getattr(cfg, "ost", False):
            # the uniform draws of every level (:3206-3207) in one launch
            a2s = [act2[-i - 1] for i in range(len(self.layers))]
            flat = torch.rand(sum(a.numel() for a in a2s), device=output.device, dtype=output.dtype)
            noise, at = [], 0
            for a in a2s:
                noise.append(flat[at:at + a.numel()].view(a.shape))
                at += a.numel()
        for i, (layer, dis, dis2) in enumerate(zip(self.layers, self.dis_layers, self.dis_layers2)):
            a1, a2 = act1[-i - 1], act2[-i - 1]
            train_branch = not restarted and not getattr(cfg, "ost", False)
            if last_sample is None:
                q_in, p_in = use(a1) + use(a2), use(a2)
            elif train_branch and nxt is not None:
                q_in, p_in = nxt                                     # formed by the previous level's DlsLevelNext launch
            elif train_branch:
                q_in, p_in = ops.DlsInputs.apply(a1, a2, last_sample, last_t)
            else:
                q_in = torch.cat([a1 + a2, last_sample], -1)
                p_in = torch.cat([a2, last_sample], -1)
            need_q = train_branch or self.training
            # the three Blocks of a level are independent: they share one launch (or run as parallel stream branches)
            res = run_blocks([dis2, layer] + ([dis] if need_q else []),
                             [p_in, use(output) if i == 0 else output] + ([q_in] if need_q else []))
            t_mean, t_lv = res[0].chunk(2, -1)
            lay_out = res[1]
            if need_q:
                mean, lv = res[2].chunk(2, -1)
            if train_branch:
                eps = noise[i] if noise is not None else torch.rand(a2.shape, device=a2.device, dtype=a2.dtype)
                # sample (:3292/:3299-3300), output (:3310) and ew * KL(t || q.detach) + cw * KL(t.detach || q) (:3305-3306);
                # all but the last level also form the next level's Block inputs (:3289-3297) in the same launch
                more = i + 1 < len(self.layers)
                if batched:
                    # the running loss rides through the level kernels (rkl <- rkl + klw inside the launch)
                    if cons_beside is True:
                        ops.join_third()                 # the sum made behind the previous level's kernel
                        cons_beside = held = None
                    rkl_t = rkl if torch.is_tensor(rkl) else None
                    last_sample, output, rkl_new, q_n, p_n = ops.DlsLevelNext.apply(
                        res[2], res[0], use(a2), eps, lay_out, float(ew), float(cw), use(act1[-i - 2]) if more else None,
                        use(act2[-i - 2]) if more else None, rkl_t)
                    rkl = rkl_new if rkl_t is not None or rkl is None else rkl + rkl_new
                    if torch.is_tensor(cons_beside):
                        held = rkl          # read on the third stream: its memory must not be handed out before the join
                        with ops.fork_third():
                            rkl = cons_beside + rkl
                        cons_beside = True
                    nxt = (q_n, p_n) if more else None
                else:
                    last_sample, output, klw = ops.DlsLevel.apply(res[2], res[0], a2, eps, lay_out, float(ew), float(cw))
                    nxt = None
                    rkl = rkl + klw + scaled(ops.cons_loss(glob(a1), glob(a2)))                      # :3307
                last_t = t_mean
            else:
                if self.training:
                    eps = noise[i] if noise is not None else torch.rand_like(t_lv)
                    z, _ = ops.reparam_kl(t_mean, t_lv, None, None, eps, mode=1, want_kl=False)    # :3346
                    last_sample = z + a2
                    _, k1 = ops.reparam_kl(t_mean, t_lv, mean.detach(), lv.detach(), None, mode=1, want_z=False)
                    rkl = rkl + ew * ops.mean(k1)                                                  # :3347
                else:
                    last_sample = t_mean + a2                                                      # :3350
                output = torch.nan_to_num(lay_out + last_sample)                                   # :3351
        if cons_beside is True:
            ops.join_third()
            held = None
        if not restarted and not getattr(cfg, "ost", False):
            dens = self.density_term(output.shape[0])
            if dens:
                rkl = rkl + dens                                                                   # :3311
        return output, rkl


def gaussian_merge(mean1, mean2, logvar1, logvar2):
    """DeN_Auto.gaussian_merge (dialog_models.py:2687-2693): product-style fusion of two Gaussians given as
    (mean, logvar); the merged variance is doubled ("Var Non decrease")."""
    var1, var2 = logvar1.exp(), logvar2.exp()
    mean = (var1 * mean2 + var2 * mean1) / (var1 + var2)
    return mean, (2 * (var1 * var2) / (var1 + var2)).log()


def gumbel_sampling(samples, gumbels=None):
    """DeN_Auto.gumbel_sampling (dialog_models.py:2674-2685): one hard-negative row index per row by Gumbel-max over
    log(1 + 1e-6 + cosine score), the diagonal pushed down by the matrix maximum.  `gumbels`: the -log(Exp(1)) draw
    of :2682 (drawn on the device when None).  No gradient (the reference passes a detached clone)."""
    with torch.no_grad():
        a = ops.L2NormRows.apply(samples.contiguous())
        sc = ops.ScaledMatmulNT.apply(a, a, 1.0)
        n = sc.size(0)
        sc = sc - torch.eye(n, device=sc.device, dtype=sc.dtype) * sc.max()
        if gumbels is None:
            gumbels = -torch.empty_like(sc).exponential_().log()
        return (torch.log(1 + 1e-6 + sc) + gumbels).max(-1)[1]


class DeNAuto(nn.Module):
    """DeN_Auto.forward(output, act1, act2, pos_embeddings, restarted) -> (z, rkl)  (dialog_models.py:2611-2764): the
    DEM / Self-Adaption-Fusion sibling of DLS_Auto, same parameter names (`layers`, `dis_layers`, `dis_layers2`,
    `y_z_info.i.{0,1}`).  Per level the posterior and prior Blocks are merged (gaussian_merge), two detached-direction KLs
    pull each towards the merged Gaussian, the sample comes from the MERGED Gaussian, and a discriminator (two Blocks +
    sigmoid) scores (prior mean, sample) pairs against Gumbel hard negatives.  `noise` / `gumbels`: the per-level random
    draws (:2658, :2682) for deterministic parity tests; drawn on the device when None.  `pos_embeddings` is unused by
    the reference and is not a parameter here."""

    def __init__(self, input_dim, output_dim, config):
        super().__init__()
        self.config = config
        self.layers, self.dis_layers, self.dis_layers2 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.y_z_info = nn.ModuleList()
        cnt = 0
        while input_dim <= output_dim:
            self.layers.append(Block(input_dim, input_dim * 2, input_dim * 2))
            k = 2 if cnt == 0 else 3
            self.dis_layers.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            self.dis_layers2.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            self.y_z_info.append(nn.Sequential(Block(input_dim * 4, input_dim * 4), Block(input_dim * 4, 1, input_dim),
                                               nn.Sigmoid()))
            input_dim *= 2
            cnt += 1
        for m in self.y_z_info.modules():                 # the discriminator scores positive and negative pairs: two
            if isinstance(m, SNLinear):                   # forwards (two power iterations, as torch's hook does) per step
                m.clone_uv = True

    @staticmethod
    def _kl(mu1, mu2, ls1, ls2):
        """gaussian_analytical_kl (:2664-2668), batch mean of the row sums (log-sigma inputs)."""
        _, k = ops.reparam_kl(mu1, ls1, mu2, ls2, None, mode=1, want_z=False)
        return ops.mean(k)

    def forward(self, output, act1, act2, restarted=False, noise=None, gumbels=None):
        cfg = self.config
        ew, cw = cfg.expand_weight, cfg.converge_weight
        rkl = 0
        last_sample = last_t = None
        train_branch = not restarted and not getattr(cfg, "ost", False)
        # config.global_negatives: the hard negative of a row is drawn among the rows of ALL ranks (dp.GatherRows; the
        # negative's gradient travels back to the rank that owns the row)
        gather, rank = dp.global_rows(cfg)
        for i, (layer, dis, dis2, m_layer) in enumerate(zip(self.layers, self.dis_layers, self.dis_layers2, self.y_z_info)):
            a1, a2 = act1[-i - 1], act2[-i - 1]
            if train_branch:
                if last_sample is None:
                    q_in, p_in = a1 + a2, a2
                else:
                    q_in, p_in = torch.cat([a1 + a2, last_sample], -1), torch.cat([a2, last_t], -1)
                res = run_blocks([dis, dis2, layer], [q_in, p_in, output])
                mean, lv = res[0].chunk(2, -1)
                t_mean, t_lv = res[1].chunk(2, -1)
                last_t = t_mean
                m_mean, m_lv = gaussian_merge(t_mean, mean, t_lv, lv)
                rkl = rkl + ew * self._kl(t_mean, m_mean.detach(), t_lv, m_lv.detach())         # :2716
                rkl = rkl + cw * self._kl(m_mean.detach(), mean, m_lv.detach(), lv)             # :2717
                eps = noise[i] if noise is not None else torch.rand_like(m_lv)
                z, _ = ops.reparam_kl(m_mean, m_lv, None, None, eps, mode=1, want_kl=False)     # :2719
                last_sample = z + a2
                if not getattr(cfg, "no_mutual_information", False):                             # :2721-2727
                    pool = last_sample if gather is None else gather(last_sample)
                    perm = gumbel_sampling(pool.detach(), None if gumbels is None else gumbels[i])
                    if gather is not None:               # this rank's rows of the global draw
                        perm = perm[rank * last_sample.shape[0]:(rank + 1) * last_sample.shape[0]]
                    pos = m_layer(torch.cat([last_t, last_sample], -1))
                    neg = m_layer(torch.cat([last_t, pool[perm]], -1))
                    rkl = rkl - torch.mean(torch.log(pos + 1e-6) + torch.log(1 - neg + 1e-6))
                output = res[2] + last_sample                                                    # :2729
            else:
                need_q = restarted or self.training
                if last_sample is None:
                    q_in, p_in = a1, a2
                else:
                    q_in, p_in = torch.cat([a1, last_sample], -1), torch.cat([a2, last_sample], -1)
                res = run_blocks([dis2, layer] + ([dis] if need_q else []), [p_in, output] + ([q_in] if need_q else []))
                t_mean, t_lv = res[0].chunk(2, -1)
                if need_q:
                    mean, lv = res[2].chunk(2, -1)
                if self.training:
                    eps = noise[i] if noise is not None else torch.rand_like(t_lv)
                    z, _ = ops.reparam_kl(t_mean, t_lv, None, None, eps, mode=1, want_kl=False)   # :2756
                    last_sample = z + a2
                    m_mean, m_lv = gaussian_merge(t_mean, mean, t_lv, lv)
                    rkl = rkl + ew * self._kl(t_mean, m_mean.detach(), t_lv, m_lv.detach())       # :2758
                    rkl = rkl + cw * self._kl(m_mean.detach(), mean, m_lv.detach(), lv)           # :2759
                else:
                    last_sample = t_mean + a2                                                    # :2761
                output = torch.nan_to_num(res[1] + last_sample)                                  # :2762
        return output, rkl
