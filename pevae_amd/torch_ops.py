"""`torch.ops.pevae_amd.*`: the inference-side entry points of the C ABI as registered torch operators (torch.library), so
that callers outside this package -- TorchScript-free serving code, torch.compile'd graphs, code that looks operators up
by name -- reach the HIP kernels without importing pevae_amd.ops.  Each operator has a fake (meta) implementation, so
shape propagation and tracing work without a GPU; the real implementation is the ctypes call of pevae_amd.ops and needs
an MI355X (there is no CPU fallback: a CPU tensor raises).

The TRAINING step keeps its torch.autograd.Function wrappers (pevae_amd.ops): their backward passes share workspaces,
gradient sinks and side streams across nodes, which a stateless operator signature cannot express; the step is replayed
from a HIP graph, where the binding's cost per call does not exist.

    import pevae_amd.torch_ops          # registers the operators
    ids, logp = torch.ops.pevae_amd.vocab_topk(hid, weight, bias, 10, 1.0, -1)
"""
from typing import Tuple

import torch

from . import ops

NAMESPACE = "pevae_amd"


@torch.library.custom_op(NAMESPACE + "::vocab_argmax", mutates_args=())
def vocab_argmax(hid: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, inv_temp: float,
                 mask_col: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pv_vocab_argmax (decoders.py:252-254): greedy symbol and its log-probability per row, no [T, V] tensor."""
    return ops.vocab_argmax(hid, weight, bias, inv_temp, mask_col)


@vocab_argmax.register_fake
def _(hid, weight, bias, inv_temp, mask_col):
    T = hid.shape[0]
    return hid.new_empty((T,), dtype=torch.int64), hid.new_empty((T,), dtype=torch.float32)


@torch.library.custom_op(NAMESPACE + "::vocab_topk", mutates_args=())
def vocab_topk(hid: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, k: int, inv_temp: float,
               mask_col: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pv_vocab_topk (decoders.py:263-270): the k best symbols per row and their log-probabilities, no [T, V] tensor."""
    return ops.vocab_topk(hid, weight, bias, k, inv_temp, mask_col)


@vocab_topk.register_fake
def _(hid, weight, bias, k, inv_temp, mask_col):
    T = hid.shape[0]
    return hid.new_empty((T, k), dtype=torch.int64), hid.new_empty((T, k), dtype=torch.float32)


@torch.library.custom_op(NAMESPACE + "::vocab_sample", mutates_args=())
def vocab_sample(hid: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, inv_temp: float, mask_col: int,
                 site: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pv_vocab_sample (decoders.py:255-256): one multinomial draw per row from the device generator, no [T, V] tensor."""
    return ops.vocab_sample(hid, weight, bias, inv_temp, mask_col, noise=None, site=site)


@vocab_sample.register_fake
def _(hid, weight, bias, inv_temp, mask_col, site):
    T = hid.shape[0]
    return hid.new_empty((T,), dtype=torch.int64), hid.new_empty((T,), dtype=torch.float32)


@torch.library.custom_op(NAMESPACE + "::greedy_decode", mutates_args=())
def greedy_decode(emb_weight: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor,
                  proj_weight: torch.Tensor, proj_bias: torch.Tensor, h0: torch.Tensor, length: int, sos_id: int,
                  inv_temp: float, mask_col: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """pv_greedy_decode (decoders.py:236-262): all `length` tokens of a greedy decode in one persistent launch."""
    return ops.greedy_decode(emb_weight, w_ih, w_hh, b_ih, b_hh, proj_weight, proj_bias, h0, length, sos_id,
                             inv_temp=inv_temp, mask_col=mask_col)


@greedy_decode.register_fake
def _(emb_weight, w_ih, w_hh, b_ih, b_hh, proj_weight, proj_bias, h0, length, sos_id, inv_temp, mask_col):
    B, H = h0.shape
    return (h0.new_empty((B, length), dtype=torch.int64), h0.new_empty((B, length), dtype=torch.float32),
            h0.new_empty((B, H), dtype=torch.float32))
