"""BaseRNN with the reference's constructor and state_dict layout (models/enc2dec/base_modules.py:9-81),
running its recurrence on the persistent HIP GRU / LSTM kernels instead of cuDNN."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops

PAD, EOS = "<pad>", "</s>"      # models/dataset/corpora.py constants used by base_modules.py:10-11


def _pick(state, i):
    """state[i] of a [layers * dirs, B, H] state; a view without a select node (whose backward is a zero fill + a copy)
    when there is only one."""
    return state.squeeze(0) if state.shape[0] == 1 else state[i]


def _stack(finals):
    return finals[0].unsqueeze(0) if len(finals) == 1 else torch.stack(finals, dim=0)


class RNNWeights(nn.Module):
    """Parameters named and initialised like torch.nn.GRU / nn.LSTM (`weight_ih_l0`, `..._reverse`), because
    reference checkpoints carry `rnn.weight_ih_l0` ... (base_modules.py:42)."""

    def __init__(self, cell, input_size, hidden_size, num_layers, bidirectional, dropout):
        super().__init__()
        self.cell, self.input_size, self.hidden_size = cell, input_size, hidden_size
        self.num_layers, self.bidirectional, self.dropout = num_layers, bidirectional, float(dropout)
        self.between_layers = ops.Dropout(self.dropout) if self.dropout > 0 and num_layers > 1 else None
        ngate = 3 if cell == "gru" else 4
        ndir = 2 if bidirectional else 1
        bound = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            width = input_size if layer == 0 else hidden_size * ndir
            for d in range(ndir):
                tail = "_l{}{}".format(layer, "_reverse" if d else "")
                shapes = {"weight_ih": (ngate * hidden_size, width), "weight_hh": (ngate * hidden_size, hidden_size),
                          "bias_ih": (ngate * hidden_size,), "bias_hh": (ngate * hidden_size,)}
                for name, shape in shapes.items():
                    self.register_parameter(name + tail, nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))

    def layer_weights(self, layer, direction):
        tail = "_l{}{}".format(layer, "_reverse" if direction else "")
        return [getattr(self, n + tail) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def forward(self, x, state=None, lengths=None, x_side=False, dx_consumer=None):
        """x_side: x was produced under ops.on_side(); dx_consumer: x is a leaf value whose gradient goes to this closure
        (ops.embed_dropout_leaf) -- single-layer unidirectional GRU only (see ops.GRULayer).
        x [B,L,in] -> (out [B,L,H*dirs], h_n [layers*dirs,B,H] (LSTM: (h_n, c_n)), hseq_bf16 of the last layer or None).
        lengths (list / tensor [B]): pack_padded_sequence semantics (encoders.py:55-65): a row's state stops at its
        length, outputs past it are zero and the time axis is cut to max(lengths).  The reverse direction of a
        bidirectional RNN then starts at each row's own last token: its input is the row's valid prefix reversed in
        place (one gather), which turns it into the forward problem with the same lengths."""
        lstm = self.cell == "lstm"
        ndir = 2 if self.bidirectional else 1
        assert dx_consumer is None or (not lstm and ndir == 1 and self.num_layers == 1 and lengths is None)
        lens = mask = None
        rev_idx = None
        if lengths is not None:
            lens_l = torch.as_tensor(lengths, dtype=torch.int64)
            x = x[:, :int(lens_l.max())]
            lens = lens_l.to(device=x.device, dtype=torch.int32)
            pos = torch.arange(x.size(1), device=x.device).unsqueeze(0)
            mask = (pos < lens.unsqueeze(1)).unsqueeze(-1).to(x.dtype)
            if self.bidirectional:                        # position j of row b <-> position len_b - 1 - j (j < len_b)
                ll = lens.to(torch.int64).unsqueeze(1)
                rev_idx = torch.where(pos < ll, ll - 1 - pos, pos).unsqueeze(-1)
        save = torch.is_grad_enabled()
        finals, cfinals, hseq = [], [], None
        for layer in range(self.num_layers):
            outs = []
            for d in range(ndir):
                w_ih, w_hh, b_ih, b_hh = self.layer_weights(layer, d)
                if d and rev_idx is not None:
                    xin = torch.gather(x, 1, rev_idx.expand(-1, -1, x.size(2)))
                else:
                    xin = x.flip(1) if d else x
                if lstm:                                      # state is (h_0, c_0), returns (h_n, c_n)
                    h0 = None if state is None else state[0][layer * ndir + d]
                    c0 = None if state is None else state[1][layer * ndir + d]
                    out, hn, cn, hs = ops.LSTMLayer.apply(xin, h0, c0, w_ih, w_hh, b_ih, b_hh, save, lens)
                    cfinals.append(cn)
                else:
                    h0 = None if state is None else _pick(state, layer * ndir + d)
                    out, hn, hs = ops.GRULayer.apply(xin, h0, w_ih, w_hh, b_ih, b_hh, save, lens,
                                                     x_side and layer == 0 and d == 0 and lengths is None,
                                                     dx_consumer if layer == 0 and d == 0 else None)
                if mask is not None:
                    out, hs = out * mask, None            # zero rows past each length (pad_packed_sequence)
                if d and rev_idx is not None:
                    outs.append(torch.gather(out, 1, rev_idx.expand(-1, -1, out.size(2))))
                else:
                    outs.append(out.flip(1) if d else out)
                finals.append(hn)
                hseq = hs if ndir == 1 else None
            x = outs[0] if ndir == 1 else torch.cat(outs, dim=-1)
            if self.between_layers is not None and self.training and layer < self.num_layers - 1:
                x = self.between_layers(x)
                hseq = None
        if lstm:
            return x, (_stack(finals), _stack(cfinals)), hseq
        return x, _stack(finals), hseq


class BaseRNN(nn.Module):
    SYM_MASK = PAD
    SYM_EOS = EOS

    KEY_ATTN_SCORE = 'attention_score'
    KEY_LENGTH = 'length'
    KEY_SEQUENCE = 'sequence'
    KEY_LATENT = 'latent'
    KEY_CLASS = 'class'
    KEY_RECOG_LATENT = 'recog_latent'
    KEY_POLICY = "policy"
    KEY_G = 'g'
    KEY_PTR_SOFTMAX = 'ptr_softmax'
    KEY_PTR_CTX = "ptr_context"

    def __init__(self, vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers, rnn_cell,
                 bidirectional):
        super().__init__()
        self.vocab_size = vocab_size
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.input_dropout_p = input_dropout_p
        self.input_dropout = ops.Dropout(p=input_dropout_p)
        cell = rnn_cell.lower()
        if cell not in ("lstm", "gru"):
            raise ValueError("Unsupported RNN Cell: {0}".format(rnn_cell))          # base_modules.py:39
        self.rnn_cell = nn.LSTM if cell == "lstm" else nn.GRU      # callers test `self.rnn_cell is nn.LSTM`
        self.dropout_p = dropout_p
        self.rnn = RNNWeights(cell, input_size, hidden_size, n_layers, bidirectional, dropout_p)
        if cell == "lstm":                                          # forget-gate bias = 1, base_modules.py:45-51
            for name, p in self.rnn.named_parameters():
                if "bias" in name:
                    n = p.size(0)
                    p.data[n // 4:n // 2].fill_(1.)

    def multinomial_sampling(self, log_probs):
        """base_modules.py:54-59 (the argument is a probability matrix despite its name)."""
        return torch.multinomial(log_probs, 1)

    def gumbel_max(self, log_probs):
        """base_modules.py:61-74."""
        u = torch.rand(log_probs.size()).to(log_probs.device)
        return torch.max(log_probs - torch.log(-torch.log(u)), dim=-1, keepdim=True)[1]

    def repeat_state(self, state, batch_size, times):
        return state.repeat(1, 1, times).view(-1, batch_size * times, self.hidden_size)

    def forward(self, *args, **kwargs):
        raise NotImplementedError()
