"""Loss modules with the reference's names and signatures (models/criterions.py:24-211) on HIP kernels.

NLLEntropy / Perplexity accept either a dense log-prob tensor (reference behaviour) or the decoder's
`DeferredLogProbs`, in which case the fused vocab GEMM + online-softmax + NLL kernel is used and the [B,L,V]
tensor is never built."""
import os

import torch
from torch.nn.modules.loss import _Loss

from . import ops
from .enc2dec.decoders import DeferredLogProbs

FUSED_HEAD = os.environ.get("PEVAE_NLL_HEAD", "1") == "1"      # ops.NllHead: 'seq' NLL and perplexity in one launch


def _token_nll(net_output, labels, padding_idx):
    """-> (sum of token NLL, number of non-pad tokens) as device scalars."""
    if isinstance(net_output, DeferredLogProbs):
        return net_output.token_nll_sums(labels, padding_idx)
    flat = net_output.reshape(-1, net_output.size(-1))
    tgt = labels.reshape(-1)
    picked = ops.GatherNLL.apply(flat, tgt, padding_idx)
    sums = ops.MaskedSum.apply(picked, tgt, padding_idx)
    return sums[0], sums[1]


class NLLEntropy(_Loss):
    """criterions.py:24-77 (weight=None path).  avg_type: 'seq' = sum / batch, 'word' = token mean, None = sum;
    'real_word' is broken in the reference (reduction=None, :46) and raises here as well."""

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None, avg_type="seq"):
        super().__init__()
        self.padding_idx = padding_idx
        self.avg_type = avg_type
        if rev_vocab is not None and key_vocab is not None:
            raise NotImplementedError("pevae_amd NLLEntropy: per-word loss weights are not supported")
        if avg_type not in ("word", "real_word", "seq", None):
            raise NotImplementedError("Unknown avg type")

    def forward(self, net_output, labels):
        batch_size = net_output.size(0)
        if self.avg_type == 'seq' and isinstance(net_output, DeferredLogProbs) and FUSED_HEAD:
            return net_output.loss_head(labels, self.padding_idx)[0]
        total, count = _token_nll(net_output, labels, self.padding_idx)
        if self.avg_type is None:
            return total
        if self.avg_type == 'seq':
            return total / batch_size
        if self.avg_type == 'word':
            return total / count
        raise ValueError("Unknown avg type")


class Perplexity(_Loss):
    """criterions.py:117-138: exp(token-mean NLL)."""

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None):
        super().__init__()
        self.padding_idx = padding_idx
        if rev_vocab is not None and key_vocab is not None:
            raise NotImplementedError("pevae_amd Perplexity: per-word loss weights are not supported")

    def forward(self, net_output, labels):
        if isinstance(net_output, DeferredLogProbs) and FUSED_HEAD:
            return net_output.loss_head(labels, self.padding_idx)[1]
        total, count = _token_nll(net_output, labels, self.padding_idx)
        return torch.exp(total / count)


class NormKLLoss(_Loss):
    """criterions.py:140-155 (logvar = log sigma^2)."""

    def __init__(self, unit_average=False):
        super().__init__()
        self.unit_average = unit_average

    def forward(self, recog_mu, recog_logvar, prior_mu, prior_logvar):
        _, kl_row = ops.reparam_kl(recog_mu, recog_logvar, prior_mu, prior_logvar, None, mode=0, want_z=False)
        if self.unit_average:
            kl_row = kl_row / recog_mu.size(1)
        return ops.mean(kl_row)


class CatKLLoss(_Loss):
    """criterions.py:157-175."""

    def forward(self, log_qy, log_py, batch_size=None, unit_average=False, average=True):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        y_kl = ops.CatRows.apply(log_qy, log_py, 0)
        if not average:
            return y_kl
        if unit_average:
            return ops.mean(y_kl)
        return ops.sum(y_kl) / batch_size


class Entropy(_Loss):
    """criterions.py:196-211."""

    def forward(self, log_qy, batch_size=None, unit_average=False):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        h_q = ops.CatRows.apply(log_qy, None, 1)
        if unit_average:
            return ops.mean(h_q)
        return ops.sum(h_q) / batch_size
