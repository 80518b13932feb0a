"""CPU restatement (PyTorch fp32) of the reference's hot-path modules.  TEST INFRASTRUCTURE.

Each class keeps the reference constructor/forward signature and state_dict keys so the
golden vectors produced from the reference (oracle/make_golden.py) can be loaded 1:1.
The arithmetic is written out explicitly (hand-rolled GRU/LSTM cells, hand-rolled
log-softmax/NLL) rather than delegated to torch.nn.GRU / F.nll_loss, so that torch's own
kernels act as an independent second opinion in tests/test_oracle_golden.py.

Reference citations are `file:line` into /root/reference (oklen/PEVAE).
"""
import math

import numpy as np
import torch
import torch.nn as nn

TEACH_FORCE = "teacher_forcing"   # models/enc2dec/decoders.py:11
TEACH_GEN = "teacher_gen"         # models/enc2dec/decoders.py:12
GEN = "gen"                       # models/enc2dec/decoders.py:13

INT, LONG, FLOAT = 0, 1, 2        # models/utils.py:23-25


class Pack(dict):
    """Attribute-style dict.  models/utils.py:28-43."""

    def __getattr__(self, name):
        return self[name]

    def add(self, **kw):
        self.update(kw)

    def copy(self):
        out = Pack()
        for k, v in self.items():
            out[k] = list(v) if type(v) is list else v
        return out


def kl_anneal_function(anneal_function, step, k, x0, warmup_step=0, warmup_value=0.0):
    """models/utils.py:244-255."""
    if step <= warmup_step:
        return warmup_value
    step = step - warmup_step
    if anneal_function == "logistic":
        return float(1.0 / (1.0 + np.exp(-k * (step - x0))))
    if anneal_function == "linear":
        return min(1.0, step / x0)
    if anneal_function == "const":
        return k
    return None


# ----------------------------------------------------------------------------------------
# recurrent cells (torch.nn.GRU / nn.LSTM semantics, which is what base_modules.py:42 builds)
# ----------------------------------------------------------------------------------------

EMULATE_BF16_MFMA = False     # True (full-size GPU tests): the products the HIP path runs on the bf16 matrix core -- the
                              # recurrences' input / hidden projections and the vocabulary projection -- round BOTH
                              # operands to bf16, forward and backward, and accumulate in fp32 like the MFMA does


def _rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _MatmulNTbf16(torch.autograd.Function):
    """y = a w^T with bf16-rounded operands; backward da = rb(dy) rb(w), dw = rb(dy)^T rb(a): what gemm_bf16.hip /
    gru.hip / vocab_fused.hip compute (their dgi / dgh / P operands are stored or formed in bf16)."""

    @staticmethod
    def forward(ctx, a, w):
        ctx.save_for_backward(a, w)
        return _rb(a) @ _rb(w).t()

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dyr = _rb(dy)
        return dyr @ _rb(w), dyr.reshape(-1, dyr.shape[-1]).t() @ _rb(a).reshape(-1, a.shape[-1])


def mm_nt(a, w):
    """a [.., K] x w [N, K]^T: fp32 as the reference computes it, or with the matrix core's operand rounding."""
    return _MatmulNTbf16.apply(a, w) if EMULATE_BF16_MFMA else a @ w.t()


def gru_cell(x_t, h, w_ih, w_hh, b_ih, b_hh):
    """One GRU step, gate order (r, z, n).  SURVEY 8(a) row a2."""
    H = h.shape[-1]
    gi = mm_nt(x_t, w_ih) + (b_ih if b_ih is not None else 0.0)
    gh = mm_nt(h, w_hh) + (b_hh if b_hh is not None else 0.0)
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1.0 - z) * n + z * h


def lstm_cell(x_t, h, c, w_ih, w_hh, b_ih, b_hh):
    """One LSTM step, gate order (i, f, g, o)."""
    H = h.shape[-1]
    g = mm_nt(x_t, w_ih) + mm_nt(h, w_hh)
    if b_ih is not None:
        g = g + b_ih + b_hh
    i = torch.sigmoid(g[..., :H])
    f = torch.sigmoid(g[..., H:2 * H])
    gg = torch.tanh(g[..., 2 * H:3 * H])
    o = torch.sigmoid(g[..., 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


class MaskedDropout(nn.Module):
    """nn.Dropout (encoders.py:54, decoders.py:175,189) with the keep mask supplied instead of drawn: y = x * keep / (1 - p)
    in training mode.  Parity tests give the same mask to the HIP modules (pevae_amd.ops.Dropout.mask)."""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        if not self.training or self.p == 0:
            return x
        return x * self.mask.to(x.dtype).view(x.shape) / (1.0 - self.p)


class RNNParams(nn.Module):
    """Parameter container with torch.nn.GRU/LSTM names (`weight_ih_l0`, `..._reverse`) and init.

    The reference keeps its RNN as `self.rnn = nn.GRU(...)` (models/enc2dec/base_modules.py:42), so
    checkpoints carry `rnn.weight_ih_l0` etc.  (checkpoint row.)
    """

    def __init__(self, cell, input_size, hidden_size, n_layers, bidirectional, dropout):
        super().__init__()
        self.cell, self.input_size, self.hidden_size = cell, input_size, hidden_size
        self.num_layers, self.bidirectional, self.dropout = n_layers, bidirectional, dropout
        gates = 3 if cell == "gru" else 4
        dirs = 2 if bidirectional else 1
        stdv = 1.0 / math.sqrt(hidden_size)
        for layer in range(n_layers):
            in_l = input_size if layer == 0 else hidden_size * dirs
            for d in range(dirs):
                sfx = "_l%d%s" % (layer, "_reverse" if d == 1 else "")
                for name, shape in (("weight_ih", (gates * hidden_size, in_l)),
                                    ("weight_hh", (gates * hidden_size, hidden_size)),
                                    ("bias_ih", (gates * hidden_size,)),
                                    ("bias_hh", (gates * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape).uniform_(-stdv, stdv))
                    self.register_parameter(name + sfx, p)

    def weights(self, layer, direction):
        sfx = "_l%d%s" % (layer, "_reverse" if direction == 1 else "")
        return tuple(getattr(self, n + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))

    use_aten = False      # True: run through stock torch.nn.GRU/LSTM kernels (what the reference calls,
                          # base_modules.py:42) sharing these Parameters -- the bench's CPU-baseline leg

    def aten_module(self):
        mod = self.__dict__.get("_aten")
        if mod is None:
            cls = nn.GRU if self.cell == "gru" else nn.LSTM
            mod = cls(self.input_size, self.hidden_size, self.num_layers, batch_first=True, dropout=self.dropout,
                      bidirectional=self.bidirectional)
            self.__dict__["_aten"] = mod
        for name, p in self.named_parameters():
            mod._parameters[name] = p
        mod.train(self.training)
        return mod


def run_rnn(params, x, state=None, lengths=None, training=False):
    """batch_first multi-layer (bi)directional GRU/LSTM over x [B,L,in].

    Returns (out [B,L,H*dirs], h_n [layers*dirs,B,H]) for GRU, (out, (h_n, c_n)) for LSTM.
    `lengths` reproduces pack_padded_sequence/pad_packed_sequence (encoders.py:55-65): state
    updates stop at each row's length and outputs past it are zero.
    """
    if params.use_aten and lengths is None:
        return params.aten_module()(x, state) if state is not None else params.aten_module()(x)
    B, L, _ = x.shape
    H, dirs = params.hidden_size, (2 if params.bidirectional else 1)
    is_lstm = params.cell == "lstm"
    if state is None:
        h0 = x.new_zeros(params.num_layers * dirs, B, H)
        c0 = x.new_zeros(params.num_layers * dirs, B, H)
    elif is_lstm:
        h0, c0 = state
    else:
        h0, c0 = state, None
    if lengths is not None:
        lens = torch.as_tensor(lengths, dtype=torch.long)
        L = int(lens.max())
        x = x[:, :L]
    layer_in = x
    hn, cn = [], []
    for layer in range(params.num_layers):
        outs_dir = []
        for d in range(dirs):
            w_ih, w_hh, b_ih, b_hh = params.weights(layer, d)
            h = h0[layer * dirs + d]
            c = c0[layer * dirs + d] if is_lstm else None
            steps = range(L - 1, -1, -1) if d == 1 else range(L)
            outs = [None] * L
            for t in steps:
                if is_lstm:
                    h2, c2 = lstm_cell(layer_in[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
                else:
                    h2 = gru_cell(layer_in[:, t], h, w_ih, w_hh, b_ih, b_hh)
                if lengths is not None:
                    live = (t < lens).unsqueeze(-1)
                    outs[t] = torch.where(live, h2, torch.zeros_like(h2))
                    h = torch.where(live, h2, h)
                    if is_lstm:
                        c = torch.where(live, c2, c)
                else:
                    outs[t] = h2
                    h = h2
                    if is_lstm:
                        c = c2
            outs_dir.append(torch.stack(outs, dim=1))
            hn.append(h)
            if is_lstm:
                cn.append(c)
        layer_in = torch.cat(outs_dir, dim=-1) if dirs == 2 else outs_dir[0]
        if params.dropout > 0 and training and layer < params.num_layers - 1:
            layer_in = torch.nn.functional.dropout(layer_in, params.dropout, True)
    h_n = torch.stack(hn, dim=0)
    if is_lstm:
        return layer_in, (h_n, torch.stack(cn, dim=0))
    return layer_in, h_n


class BaseRNN(nn.Module):
    """models/enc2dec/base_modules.py:9-81."""
    SYM_MASK = "<pad>"
    SYM_EOS = "</s>"
    KEY_ATTN_SCORE = "attention_score"
    KEY_LENGTH = "length"
    KEY_SEQUENCE = "sequence"
    KEY_LATENT = "latent"
    KEY_CLASS = "class"
    KEY_RECOG_LATENT = "recog_latent"
    KEY_POLICY = "policy"
    KEY_G = "g"
    KEY_PTR_SOFTMAX = "ptr_softmax"
    KEY_PTR_CTX = "ptr_context"

    def __init__(self, vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers,
                 rnn_cell, bidirectional):
        super().__init__()
        self.vocab_size, self.input_size, self.hidden_size = vocab_size, input_size, hidden_size
        self.n_layers, self.input_dropout_p, self.dropout_p = n_layers, input_dropout_p, dropout_p
        self.input_dropout = nn.Dropout(p=input_dropout_p)
        cell = rnn_cell.lower()
        if cell not in ("lstm", "gru"):
            raise ValueError("Unsupported RNN Cell: {0}".format(rnn_cell))   # base_modules.py:39
        self.rnn_cell_name = cell
        self.rnn = RNNParams(cell, input_size, hidden_size, n_layers, bidirectional, dropout_p)
        if cell == "lstm":      # forget-gate bias := 1 (base_modules.py:45-51)
            for name, p in self.rnn.named_parameters():
                if "bias" in name:
                    n = p.size(0)
                    p.data[n // 4:n // 2].fill_(1.0)

    def multinomial_sampling(self, probs):          # base_modules.py:54-59
        return torch.multinomial(probs, 1)

    def repeat_state(self, state, batch_size, times):   # base_modules.py:76-78
        return state.repeat(1, 1, times).view(-1, batch_size * times, self.hidden_size)


class EncoderRNN(BaseRNN):
    """models/enc2dec/encoders.py:7-66."""

    def __init__(self, input_size, hidden_size, input_dropout_p=0, dropout_p=0, n_layers=1,
                 rnn_cell="gru", variable_lengths=False, bidirection=False):
        super().__init__(-1, input_size, hidden_size, input_dropout_p, dropout_p, n_layers,
                         rnn_cell, bidirection)
        self.variable_lengths = variable_lengths
        self.output_size = hidden_size * 2 if bidirection else hidden_size

    def forward(self, input_var, input_lengths=None, init_state=None):
        emb = self.input_dropout(input_var)                                    # encoders.py:54
        lens = input_lengths if self.variable_lengths else None                # encoders.py:55-58
        return run_rnn(self.rnn, emb, init_state, lens, self.training)         # encoders.py:59-66


class RnnUttEncoder(nn.Module):
    """models/enc2dec/encoders.py:69-135: word embedding (+ one feature column per utterance) -> EncoderRNN over every
    utterance of every context -> the utterance vector: softmax-weighted sum of the states (use_attn, :117-122) or the
    last state(s) (:123-126)."""

    def __init__(self, utt_cell_size, dropout, rnn_cell="gru", bidirection=True, use_attn=False, embedding=None,
                 vocab_size=None, embed_dim=None, feat_size=0):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, embed_dim) if embedding is None else embedding       # :77-82
        self.embed_size = self.embedding.embedding_dim
        self.rnn = EncoderRNN(self.embed_size + feat_size, utt_cell_size, 0.0, dropout, rnn_cell=rnn_cell,
                              variable_lengths=False, bidirection=bidirection)                           # :84-87
        self.output_size = utt_cell_size * (2 if bidirection else 1)
        self.use_attn = use_attn
        if use_attn:                                                                                     # :93-96
            self.key_w = nn.Linear(self.output_size, utt_cell_size)
            self.query = nn.Linear(utt_cell_size, 1)

    def forward(self, utterances, feats=None, init_state=None, return_all=False):
        B, n_ctx, n_tok = (int(s) for s in utterances.size())
        if init_state is not None:
            init_state = init_state.repeat(1, n_ctx, 1)                                                  # :104-105
        words = self.embedding(utterances.view(-1, n_tok))                                               # :108-109
        if feats is not None:                                                                            # :111-114
            words = torch.cat([words, feats.view(-1, 1).unsqueeze(1).repeat(1, n_tok, 1)], dim=2)
        enc_outs, enc_last = self.rnn(words, init_state=init_state)                                      # :116
        if self.use_attn:                                                                                # :118-122
            score = self.query(torch.tanh(self.key_w(enc_outs))).squeeze(2)
            e = (score - score.max(dim=-1, keepdim=True).values).exp()
            attn = (e / e.sum(dim=-1, keepdim=True)).unsqueeze(2)
            utt = (attn * enc_outs).sum(dim=1)
        else:                                                                                            # :124-126
            attn = None
            utt = enc_last.transpose(0, 1).contiguous().view(-1, self.output_size)
        utt = utt.view(B, n_ctx, self.output_size)                                                       # :128
        return (utt, enc_outs, enc_last, attn) if return_all else utt


class Attention(nn.Module):
    """models/enc2dec/decoders.py:15-118 ('dot' | 'general' | 'cat')."""

    def __init__(self, dec_size, attn_size, mode, project=False):
        super().__init__()
        self.mask, self.mode, self.attn_size, self.dec_size = None, mode, attn_size, dec_size
        self.linear_out = nn.Linear(dec_size + attn_size, dec_size) if project else None
        if mode == "general":
            self.attn_w = nn.Linear(dec_size, attn_size)
        elif mode == "cat":
            self.dec_w = nn.Linear(dec_size, dec_size)
            self.attn_w = nn.Linear(attn_size, dec_size)
            self.query_w = nn.Linear(dec_size, 1)

    def forward(self, output, context):
        if context.size(0) != output.size(0):                                  # decoders.py:76-77
            context = context.repeat(output.size(0) // context.size(0), 1, 1)
        B, Lin = output.size(0), context.size(1)
        if self.mode == "dot":
            score = output @ context.transpose(1, 2)
        elif self.mode == "general":
            score = self.attn_w(output) @ context.transpose(1, 2)
        elif self.mode == "cat":                                               # decoders.py:89-95
            k = self.attn_w(context).unsqueeze(1)            # [B,1,Lin,D]
            q = self.dec_w(output).unsqueeze(2)              # [B,Lout,1,D]
            score = self.query_w(torch.tanh(k + q)).squeeze(-1)
        else:
            raise ValueError("Unknown attention")
        if self.mask is not None:
            score = score.masked_fill(self.mask, -float("inf"))
        smax = score - score.max(dim=-1, keepdim=True).values
        attn = smax.exp() / smax.exp().sum(dim=-1, keepdim=True)               # decoders.py:103
        mix = attn @ context
        combined = torch.cat((mix, output), dim=2)
        if self.linear_out is None:
            return combined, attn
        out = torch.tanh(self.linear_out(combined.view(-1, self.dec_size + self.attn_size)))
        return out.view(B, -1, self.dec_size), attn


def log_softmax_rows(logits):
    m = logits.max(dim=-1, keepdim=True).values
    return logits - m - (logits - m).exp().sum(dim=-1, keepdim=True).log()


class DecoderRNN(BaseRNN):
    """models/enc2dec/decoders.py:121-337.

    Deviation (documented in DESIGN.md): the reference forces `use_attention = False` right before
    constructing `Attention` (decoders.py:154) while keeping `self.use_attention`, so passing
    `attn_context` raises AttributeError.  Here Attention IS constructed when requested.
    `unk_mask` is built on the parameter device instead of `.cuda()` (decoders.py:144).
    """

    def __init__(self, vocab_size, max_len, input_size, hidden_size, sos_id, eos_id, unk_id,
                 n_layers=1, rnn_cell="lstm", input_dropout_p=0, dropout_p=0, use_attention=False,
                 attn_mode="cat", attn_size=None, use_gpu=True, embedding=None, output_size=None,
                 tie_output_embed=False, init_range=0.1, mask_unk=False, softmax_temperature=1.0):
        super().__init__(vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers,
                         rnn_cell, False)
        self.output_size = vocab_size if output_size is None else output_size
        self.max_length, self.use_attention = max_len, use_attention
        self.eos_id, self.sos_id, self.unk_id = eos_id, sos_id, unk_id
        self.use_gpu, self.softmax_temperature, self.mask_unk = use_gpu, softmax_temperature, mask_unk
        onehot = torch.zeros(self.output_size)
        onehot[unk_id] = -np.inf
        self.register_buffer("unk_mask", onehot, persistent=False)
        self.embedding = nn.Embedding(vocab_size, input_size) if embedding is None else embedding
        if use_attention:
            self.attention = Attention(hidden_size, attn_size, attn_mode, project=True)
        self.tie_output_embed = tie_output_embed
        if not tie_output_embed:
            self.project = nn.Linear(hidden_size, self.output_size)
        self.output_dropout = nn.Dropout(p=input_dropout_p)                    # decoders.py:166
        for p in self.parameters():                                            # decoders.py:167-168
            nn.init.uniform_(p.data, -init_range, init_range)

    def _project(self, flat):
        if self.tie_output_embed:                                              # decoders.py:159-160
            return mm_nt(flat, self.embedding.weight)
        return mm_nt(flat, self.project.weight) + self.project.bias

    def forward_step(self, input_var, hidden, encoder_outputs, latent_variable=None):
        B, Lout = input_var.size(0), input_var.size(1)
        emb = self.input_dropout(self.embedding(input_var))                    # decoders.py:174-175
        if latent_variable is not None:                                        # decoders.py:177-180
            emb = torch.cat([emb, latent_variable.unsqueeze(1).repeat(1, Lout, 1)], dim=-1)
        out, hidden = run_rnn(self.rnn, emb, hidden, None, self.training)      # decoders.py:182
        attn = None
        if self.use_attention and encoder_outputs is not None:                 # decoders.py:185-186
            out, attn = self.attention(out, encoder_outputs)
        out = self.output_dropout(out.contiguous())
        logits = self._project(out.view(-1, self.hidden_size)) / self.softmax_temperature
        return log_softmax_rows(logits).view(B, Lout, -1), hidden, attn        # decoders.py:191-194

    def forward(self, batch_size, inputs=None, init_state=None, attn_context=None,
                mode=TEACH_FORCE, gen_type="greedy", beam_size=4, latent_variable=None,
                feature_input=None):
        ret = dict()
        self.max_length = inputs.size(1)                                       # decoders.py:201
        if self.use_attention:
            ret[self.KEY_ATTN_SCORE] = []
        if mode == GEN:
            inputs = None
        if gen_type != "beam":
            beam_size = 1
        if inputs is not None:
            dec_in = inputs
        else:                                                                  # decoders.py:216-219
            dev = self.embedding.weight.device
            dec_in = torch.full((batch_size * beam_size, 1), self.sos_id, dtype=torch.long, device=dev)
        is_lstm = self.rnn_cell_name == "lstm"
        if mode == GEN and gen_type == "beam":                                 # decoders.py:221-229
            if is_lstm:
                hid = tuple(self.repeat_state(s, batch_size, beam_size) for s in init_state)
            else:
                hid = self.repeat_state(init_state, batch_size, beam_size)
        elif init_state is None:
            hid = None
        elif is_lstm:
            hid = tuple(s.repeat(self.n_layers, 1, 1) for s in init_state)
        else:
            hid = init_state.repeat(self.n_layers, 1, 1)                       # decoders.py:237

        symbols_seq, back_ptrs = [], []
        lengths = np.array([self.max_length] * batch_size * beam_size)         # decoders.py:242

        if mode == TEACH_FORCE:                                                # decoders.py:286-292
            dec_out, hid, _ = self.forward_step(dec_in, hid, attn_context, latent_variable)
        else:
            step_outs, cum = [], None
            for di in range(self.max_length):                                  # decoders.py:298-303
                so, hid, sattn = self.forward_step(dec_in, hid, attn_context, latent_variable)
                step_outs.append(so)
                sl = so.squeeze(1)
                if self.mask_unk:
                    sl = sl + self.unk_mask
                if self.use_attention:
                    ret[self.KEY_ATTN_SCORE].append(sattn)
                if gen_type == "greedy":
                    sym = sl.topk(1)[1]
                elif gen_type == "sample":
                    noise = getattr(self, "sample_noise", None)     # Exp(1) draws [L, B, V] of torch.multinomial's sampler
                    sym = (self.multinomial_sampling(torch.exp(sl)) if noise is None
                           else (torch.exp(sl) / noise[di]).argmax(-1, keepdim=True))
                elif gen_type == "beam":                                       # decoders.py:257-270
                    if di == 0:
                        score = sl.view(batch_size, -1)[:, :self.output_size]
                    else:
                        score = (cum + sl).view(batch_size, -1)
                    top_v, top_id = score.topk(beam_size)
                    back_ptrs.append(torch.div(top_id, self.output_size, rounding_mode="floor").view(-1, 1))
                    sym = top_id.fmod(self.output_size).view(-1, 1)
                    cum = top_v.view(-1, 1)
                else:
                    raise ValueError("Unsupported decoding mode")              # decoders.py:272
                symbols_seq.append(sym)
                eos = sym.eq(self.eos_id).cpu().view(-1).numpy()               # decoders.py:276-280
                upd = ((lengths > di) & eos) != 0
                lengths[upd] = len(symbols_seq)
                dec_in = sym
            dec_out = torch.cat(step_outs, dim=1)
            if gen_type == "beam":                                             # decoders.py:307-328
                best = cum.view(-1, beam_size).topk(1)[1].view(-1)
                final = []
                for sym, bp in zip(symbols_seq[::-1], back_ptrs[::-1]):
                    s2, b2 = sym.view(-1, beam_size), bp.view(-1, beam_size)
                    rows = torch.arange(batch_size)
                    final.append(s2[rows, best].unsqueeze(1))
                    best = b2[rows, best]
                symbols_seq = final[::-1]
        ret[self.KEY_SEQUENCE] = symbols_seq
        ret[self.KEY_LENGTH] = lengths.tolist()
        return dec_out, hid, ret


# ----------------------------------------------------------------------------------------
# connectors  (models/nn_lib.py)
# ----------------------------------------------------------------------------------------

class LinearConnector(nn.Module):
    """models/nn_lib.py:136-162."""

    def __init__(self, input_size, output_size, is_lstm, has_bias=True):
        super().__init__()
        if is_lstm:
            self.linear_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.linear_c = nn.Linear(input_size, output_size, bias=has_bias)
        else:
            self.linear = nn.Linear(input_size, output_size, bias=has_bias)
        self.is_lstm = is_lstm

    def forward(self, inputs):
        if self.is_lstm:
            return self.linear_h(inputs).unsqueeze(0), self.linear_c(inputs).unsqueeze(0)
        return self.linear(inputs).unsqueeze(0)

    def get_w(self):
        return self.linear_h.weight if self.is_lstm else self.linear.weight


class Hidden2Gaussian(nn.Module):
    """models/nn_lib.py:188-223."""

    def __init__(self, input_size, output_size, is_lstm=False, has_bias=True):
        super().__init__()
        if is_lstm:
            self.mu_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar_h = nn.Linear(input_size, output_size, bias=has_bias)
            self.mu_c = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar_c = nn.Linear(input_size, output_size, bias=has_bias)
        else:
            self.mu = nn.Linear(input_size, output_size, bias=has_bias)
            self.logvar = nn.Linear(input_size, output_size, bias=has_bias)
        self.is_lstm = is_lstm

    def forward(self, inputs):
        if self.is_lstm:
            h, c = inputs
            if h.dim() == 3:
                h, c = h.squeeze(0), c.squeeze(0)
            return self.mu_h(h) + self.mu_c(c), self.logvar_h(h) + self.logvar_c(c)
        if inputs.dim() == 3:
            inputs = inputs.squeeze(0)
        x = inputs.squeeze(0)                                                  # nn_lib.py:221
        return self.mu(x), self.logvar(x)


def gumbel_noise_from_uniform(u, eps=1e-20):
    """nn_lib.py:230-234 with the uniform draw `u` passed in."""
    return -torch.log(-torch.log(u + eps) + eps)


class GumbelConnector(nn.Module):
    """models/nn_lib.py:226-265.  `u` (optional) injects the uniform noise; the reference draws it
    with CPU `torch.rand(logits.size())` (nn_lib.py:231), which `u=None` reproduces."""

    def sample_gumbel(self, logits, use_gpu, eps=1e-20, u=None):
        if u is None:
            u = torch.rand(logits.size())
        return gumbel_noise_from_uniform(u, eps).to(logits.device, torch.float32)

    def gumbel_softmax_sample(self, logits, temperature, use_gpu, u=None):
        y = (logits + self.sample_gumbel(logits, use_gpu, u=u)) / temperature
        y = y - y.max(dim=-1, keepdim=True).values
        return y.exp() / y.exp().sum(dim=-1, keepdim=True)

    def forward(self, logits, temperature, use_gpu, hard=False, return_max_id=False,
                straight_through=False, u=None):
        y = self.gumbel_softmax_sample(logits, temperature, use_gpu, u=u)
        y_hard = y.max(dim=-1, keepdim=True)[1]
        if straight_through or hard:                                           # nn_lib.py:253-256
            onehot = torch.zeros_like(y).scatter_(1, y_hard, 1.0)
            y = (onehot - y).detach() + y
        return (y, y_hard) if return_max_id else y


class GreedyConnector(nn.Module):
    """models/nn_lib.py:268-286."""

    def forward(self, logits, use_gpu, return_max_id=False):
        y_hard = logits.max(dim=1, keepdim=True)[1]
        y = torch.zeros_like(logits).scatter_(1, y_hard, 1.0)
        return (y, y_hard) if return_max_id else y


# ----------------------------------------------------------------------------------------
# criterions  (models/criterions.py)
# ----------------------------------------------------------------------------------------

def _masked_token_nll(net_output, labels, padding_idx):
    """-log p[label] per token with pad tokens zeroed; returns (nll_tok [T], mask [T])."""
    flat = net_output.reshape(-1, net_output.size(-1))
    tgt = labels.reshape(-1)
    mask = tgt.ne(padding_idx)
    picked = flat.gather(1, tgt.clamp_min(0).unsqueeze(1)).squeeze(1)
    return -(picked * mask.to(picked.dtype)), mask


class NLLEntropy(nn.Module):
    """models/criterions.py:24-77 (weight=None path; 'real_word' is broken in the reference, :46)."""

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None, avg_type="seq"):
        super().__init__()
        self.padding_idx, self.avg_type = padding_idx, avg_type
        if avg_type not in ("word", "real_word", "seq", None):
            raise NotImplementedError("Unknown avg type")                      # criterions.py:52

    def forward(self, net_output, labels):
        nll, mask = _masked_token_nll(net_output, labels, self.padding_idx)
        if self.avg_type is None:
            return nll.sum()
        if self.avg_type == "seq":                                             # criterions.py:62-64
            return nll.sum() / net_output.size(0)
        if self.avg_type == "word":                                            # criterions.py:43-44,72-73
            return nll.sum() / mask.sum().to(nll.dtype)
        raise ValueError("Unknown avg type")


class Perplexity(nn.Module):
    """models/criterions.py:117-138: exp(token-mean NLL)."""

    def __init__(self, padding_idx, config, rev_vocab=None, key_vocab=None):
        super().__init__()
        self.padding_idx = padding_idx

    def forward(self, net_output, labels):
        nll, mask = _masked_token_nll(net_output, labels, self.padding_idx)
        return torch.exp(nll.sum() / mask.sum().to(nll.dtype))


class NormKLLoss(nn.Module):
    """models/criterions.py:140-155 (logvar = log sigma^2)."""

    def __init__(self, unit_average=False):
        super().__init__()
        self.unit_average = unit_average

    def forward(self, recog_mu, recog_logvar, prior_mu, prior_logvar):
        inv_p = torch.exp(-prior_logvar)
        t = 1.0 + (recog_logvar - prior_logvar) - (prior_mu - recog_mu) ** 2 * inv_p \
            - torch.exp(recog_logvar) * inv_p
        red = t.mean(dim=1) if self.unit_average else t.sum(dim=1)
        return (-0.5 * red).mean()


class CatKLLoss(nn.Module):
    """models/criterions.py:157-175."""

    def forward(self, log_qy, log_py, batch_size=None, unit_average=False, average=True):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        y_kl = (log_qy.exp() * (log_qy - log_py)).sum(dim=-1)
        if not average:
            return y_kl
        return y_kl.mean() if unit_average else y_kl.sum() / batch_size


class Entropy(nn.Module):
    """models/criterions.py:196-211."""

    def forward(self, log_qy, batch_size=None, unit_average=False):
        if log_qy.dim() > 2:
            log_qy = log_qy.squeeze()
        h_q = (-log_qy * log_qy.exp()).sum(dim=1)
        return h_q.mean() if unit_average else h_q.sum() / batch_size


# ----------------------------------------------------------------------------------------
# hot-path pieces of models/dialog_models.py
# ----------------------------------------------------------------------------------------

def reparameterization(mu, logvar, eps):
    """GMMBase.reparameterization dialog_models.py:191-203: std = exp(logvar) (NOT exp(0.5*logvar));
    hash_rand :323-325 and DLS_Auto.dis_reparameterization :3206-3207 have the same form (eps given /
    eps ~ U[0,1))."""
    return mu + torch.exp(logvar) * eps


def gaussian_analytical_kl(mu1, mu2, logsigma1, logsigma2):
    """dialog_models.py:3215-3217 / :3482-3484 (inputs are log sigma)."""
    t = -0.5 + logsigma2 - logsigma1 + 0.5 * (torch.exp(2.0 * logsigma1) + (mu1 - mu2) ** 2) \
        / torch.exp(2.0 * logsigma2)
    return t.sum(dim=1).mean(dim=0)


def gaussian_analytical_kl_standard(mu1, logsigma1):
    """dialog_models.py:3220-3221."""
    return (0.5 - logsigma1 + 0.5 * (torch.exp(2.0 * logsigma1) + mu1 ** 2)).sum(dim=1).mean(dim=0)


def gaussian_merge(mean1, mean2, logvar1, logvar2):
    """dialog_models.py:3240-3244."""
    v1, v2 = logvar1.exp(), logvar2.exp()
    return (v1 * mean2 + v2 * mean1) / (v1 + v2), (2.0 * v1 * v2 / (v1 + v2)).log()


def cons_loss(v1, v2, batch_arrange):
    """InfoNCE, dialog_models.py:3266-3272."""
    a = v1 / v1.norm(dim=1, keepdim=True)
    b = v2 / v2.norm(dim=1, keepdim=True)
    s = a @ b.t() * 1.1
    rows = -log_softmax_rows(s).gather(1, batch_arrange.unsqueeze(1)).mean()
    cols = -log_softmax_rows(s.t()).gather(1, batch_arrange.unsqueeze(1)).mean()
    return (rows + cols) / 2


class SNLinear(nn.Module):
    """`spectral_norm(nn.Linear)` as torch.nn.utils.spectral_norm implements it (hook form, 1 power
    iteration per training forward, eps=1e-12): parameters `weight_orig`, `bias`; buffers `weight_u`,
    `weight_v`.  Used by Block, dialog_models.py:258-263."""

    def __init__(self, in_f, out_f, eps=1e-12):
        super().__init__()
        lin = nn.Linear(in_f, out_f)
        self.weight_orig = nn.Parameter(lin.weight.data.clone())
        self.bias = nn.Parameter(lin.bias.data.clone())
        self.eps = eps
        u = torch.nn.functional.normalize(torch.randn(out_f), dim=0, eps=eps)
        v = torch.nn.functional.normalize(torch.randn(in_f), dim=0, eps=eps)
        self.register_buffer("weight_u", u)
        self.register_buffer("weight_v", v)

    def effective_weight(self):
        w = self.weight_orig
        u, v = self.weight_u, self.weight_v
        if self.training:
            with torch.no_grad():
                v = torch.nn.functional.normalize(w.t() @ u, dim=0, eps=self.eps)
                u = torch.nn.functional.normalize(w @ v, dim=0, eps=self.eps)
                self.weight_u.copy_(u)
                self.weight_v.copy_(v)
            u, v = u.clone(), v.clone()
        sigma = torch.dot(u, w @ v)
        return w / sigma

    def forward(self, x):
        return x @ self.effective_weight().t() + self.bias


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


class Block(nn.Module):
    """dialog_models.py:253-271: 4 spectral-norm linears with GELU pre-activation, residual +
    LayerNorm, then a spectral-norm `translate`."""

    def __init__(self, input_dim, output_dim, mid_dim=None):
        super().__init__()
        mid = input_dim // 2 if mid_dim is None else mid_dim
        self.c1 = SNLinear(input_dim, input_dim)
        self.c2 = SNLinear(input_dim, mid)
        self.c3 = SNLinear(mid, mid)
        self.c4 = SNLinear(mid, input_dim)
        self.layer_norm = nn.LayerNorm(input_dim)
        self.translate = SNLinear(input_dim, output_dim)

    def forward(self, x):
        y = self.c1(gelu(x))
        y = self.c2(gelu(y))
        y = self.c3(gelu(y))
        y = self.c4(gelu(y))
        s = x + y
        mu = s.mean(dim=-1, keepdim=True)
        var = ((s - mu) ** 2).mean(dim=-1, keepdim=True)
        n = (s - mu) / torch.sqrt(var + self.layer_norm.eps)
        return self.translate(n * self.layer_norm.weight + self.layer_norm.bias)


class DlsPyramid(nn.Module):
    """DlsRec / DlsPri, dialog_models.py:1835-1870: halving Blocks while input_dim >= output_dim; the
    activations *entering* each Block are returned.  `wrap=True` reproduces DlsPri's nn.Sequential
    wrapper (state_dict keys `layers.i.0.*` instead of `layers.i.*`).  The reference indexes
    `output[:, 0]` (a [CLS] row of a transformer output); callers here pass [B,1,D] or [B,D]."""

    def __init__(self, input_dim, output_dim, config=None, wrap=False):
        super().__init__()
        self.layers = nn.ModuleList()
        d = input_dim
        while d >= output_dim:
            blk = Block(d, d // 2, d // 2)
            self.layers.append(nn.Sequential(blk) if wrap else blk)
            d //= 2

    def forward(self, output):
        if output.dim() == 3:
            output = output[:, 0]
        acts = []
        for layer in self.layers:
            acts.append(output)
            output = layer(output)
        return output, acts


def density_estimate(mean, logvar, sample, n_batch):
    """DLS_Auto.density_estimate + _density_estimate, dialog_models.py:3227-3236, with the standard-normal log density of
    gau_list[i] (:3198, :3203) written out: for every pair (a, b) the sample of row a is standardised by the Gaussian of
    row b, log N(0, 1) is summed over the features and over b, log(n_batch * B) is subtracted, and the mean over a is
    taken.  `n_batch` is the reference's self.n_batch (:3169 sets the Python int -1, on which torch.log raises: the
    term only runs with n_batch a tensor, see oracle/make_golden.py gen_dls_auto)."""
    var = logvar.exp()
    rs = sample.expand(sample.size(0), sample.size(0), sample.size(1)).transpose(0, 1)      # rs[a][b] = sample[a]
    x = (rs - mean) / var
    log_prob = -0.5 * x * x - 0.5 * math.log(2 * math.pi)
    prob = log_prob.sum(-1).sum(-1) - torch.log(n_batch * var.size(0))
    return prob.mean()


class DLSAuto(nn.Module):
    """DLS_Auto.forward, dialog_models.py:3274-3352 (train branch :3285-3311, restarted / ost branch :3312-3351),
    restated on the oracle Block, including the density-estimate regulariser (:3232-3236, :3308, :3311) when
    config.gir_weight != 0 (it needs `self.n_batch` as a tensor: the reference's Python int -1 makes torch.log raise).
    `batch_arrange` is arange(actual batch) instead of arange(config.batch_size) (:3186).  The uniform draws of
    dis_reparameterization (:3206-3207) are passed in as `noise` (one [B, d] tensor per level).  Pinned by
    tests/golden/dls_auto_*.npz, generated from the reference class itself."""

    def __init__(self, input_dim, output_dim, config):
        super().__init__()
        self.config = config
        self.layers, self.dis_layers, self.dis_layers2 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        cnt = 0
        while input_dim <= output_dim:
            self.layers.append(Block(input_dim, input_dim * 2, input_dim * 2))
            k = 2 if cnt == 0 else 3
            self.dis_layers.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            self.dis_layers2.append(Block(input_dim * k, input_dim * 4, input_dim * 4))
            input_dim *= 2
            cnt += 1

    def forward(self, output, rec_output, act1, act2, restarted=False, noise=None):
        cfg = self.config
        cip, ew, cw = getattr(cfg, "cip_weight", 1.0), cfg.expand_weight, cfg.converge_weight
        arange = torch.arange(output.size(0))
        rkl = cons_loss(output, rec_output, arange) * cip
        last_sample = last_t = None
        train_branch = not restarted and not getattr(cfg, "ost", False)
        gir = getattr(cfg, "gir_weight", 0.0)
        n_batch = getattr(self, "n_batch", None)
        qgroup, amean, alogvar, asample = 0, None, None, None                                      # :3280-3283
        for i, (layer, dis, dis2) in enumerate(zip(self.layers, self.dis_layers, self.dis_layers2)):
            a1, a2 = act1[-i - 1], act2[-i - 1]
            if last_sample is None:
                q_in, p_in = a1 + a2, a2
            else:
                q_in = torch.cat([a1 + a2, last_sample], -1)
                p_in = torch.cat([a2, last_t if train_branch else last_sample], -1)
            t_mean, t_lv = dis2(p_in).chunk(2, -1)
            if train_branch or self.training:
                mean, lv = dis(q_in).chunk(2, -1)
            if train_branch:
                eps = noise[i] if noise is not None else torch.rand_like(lv)
                last_sample = reparameterization(mean, lv, eps) + a2
                last_t = t_mean
                rkl = rkl + ew * gaussian_analytical_kl(t_mean, mean.detach(), t_lv, lv.detach())
                rkl = rkl + cw * gaussian_analytical_kl(t_mean.detach(), mean, t_lv.detach(), lv)
                rkl = rkl + cons_loss(a1, a2, arange) * cip
                if gir != 0:
                    cat = (lambda u, v: v if u is None else torch.cat([u, v], -1))
                    amean, alogvar, asample = cat(amean, t_mean), cat(alogvar, t_lv), cat(asample, last_sample)   # :3293-3302
                    qgroup = qgroup + density_estimate(t_mean, t_lv, last_sample, n_batch)          # :3308
                output = layer(output) + last_sample
            else:
                if self.training:
                    eps = noise[i] if noise is not None else torch.rand_like(t_lv)
                    last_sample = reparameterization(t_mean, t_lv, eps) + a2
                    rkl = rkl + ew * gaussian_analytical_kl(t_mean, mean.detach(), t_lv, lv.detach())
                else:
                    last_sample = t_mean + a2
                output = torch.nan_to_num(layer(output) + last_sample)
        if train_branch and gir != 0:
            rkl = rkl + torch.abs((qgroup - density_estimate(amean, alogvar, asample, n_batch)) * gir)   # :3311
        return output, rkl


def gumbel_sampling(samples, gumbels):
    """DeN_Auto.gumbel_sampling, dialog_models.py:2674-2685: hard-negative indices.  Row-normalise, cosine scores
    with the diagonal pushed down by the matrix maximum, log(1 + 1e-6 + score), Gumbel-max over each row.  The reference
    draws `gumbels = -log(Exp(1))` inside (:2682); here the draw is an argument."""
    ratio = torch.sqrt(torch.sum(samples * samples, -1))
    samples = samples / ratio.unsqueeze(-1)
    sc = samples @ samples.transpose(0, 1)
    dig = torch.diag(torch.ones_like(ratio) * torch.max(sc))
    scores = torch.log(1 + 1e-6 + sc - dig)
    return (scores + gumbels).max(-1)[1]


class DeNAuto(nn.Module):
    """DeN_Auto, dialog_models.py:2611-2764: the DEM / Self-Adaption-Fusion sibling of DLS_Auto.  Per level: posterior
    and prior Blocks, gaussian_merge of the two (:2687-2693), two detached-direction KLs against the merged Gaussian
    (:2716-2717), uniform reparameterisation of the MERGED Gaussian (:2719) and the discriminator term
    -mean(log(pos + 1e-6) + log(1 - neg + 1e-6)) with Gumbel hard negatives (:2721-2727).  `noise`: per level the
    uniform draw of :2658 ([B, 2d]); `gumbels`: per level the -log(Exp(1)) draw of :2682 ([B, B]).
    The restarted / ost branch (:2730-2762) is restated with `mean, logvar` computed whenever they are used (the
    reference reads them unbound when ost is set without restarted)."""

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

    def forward(self, output, act1, act2, restarted=False, noise=None, gumbels=None):
        cfg = self.config
        ew, cw = cfg.expand_weight, cfg.converge_weight
        rkl = 0
        last_sample = last_t = None
        train_branch = not restarted and not getattr(cfg, "ost", False)
        for i, (layer, dis, dis2, m_layer) in enumerate(zip(self.layers, self.dis_layers, self.dis_layers2, self.y_z_info)):
            a1, a2 = act1[-i - 1], act2[-i - 1]
            if train_branch:
                if last_sample is None:
                    q_in, p_in = a1 + a2, a2
                else:
                    q_in, p_in = torch.cat([a1 + a2, last_sample], -1), torch.cat([a2, last_t], -1)
                mean, lv = dis(q_in).chunk(2, -1)
                t_mean, t_lv = dis2(p_in).chunk(2, -1)
                last_t = t_mean
                m_mean, m_lv = gaussian_merge(t_mean, mean, t_lv, lv)
                rkl = rkl + ew * gaussian_analytical_kl(t_mean, m_mean.detach(), t_lv, m_lv.detach())
                rkl = rkl + cw * gaussian_analytical_kl(m_mean.detach(), mean, m_lv.detach(), lv)
                eps = noise[i] if noise is not None else torch.rand_like(m_lv)
                last_sample = reparameterization(m_mean, m_lv, eps) + a2
                if not getattr(cfg, "no_mutual_information", False):
                    g = gumbels[i] if gumbels is not None else -torch.empty(a2.size(0), a2.size(0)).exponential_().log()
                    perm = gumbel_sampling(last_sample.detach().clone(), g)
                    pos = m_layer(torch.cat([last_t, last_sample], -1))
                    neg = m_layer(torch.cat([last_t, last_sample[perm]], -1))
                    rkl = rkl - torch.mean(torch.log(pos + 1e-6) + torch.log(1 - neg + 1e-6))
                output = layer(output) + last_sample
            else:
                need_q = restarted or self.training
                if last_sample is None:
                    q_in, p_in = a1, a2
                else:
                    q_in, p_in = torch.cat([a1, last_sample], -1), torch.cat([a2, last_sample], -1)
                t_mean, t_lv = dis2(p_in).chunk(2, -1)
                if need_q:
                    mean, lv = dis(q_in).chunk(2, -1)
                if self.training:
                    eps = noise[i] if noise is not None else torch.rand_like(t_lv)
                    last_sample = reparameterization(t_mean, t_lv, eps) + a2
                    m_mean, m_lv = gaussian_merge(t_mean, mean, t_lv, lv)
                    rkl = rkl + ew * gaussian_analytical_kl(t_mean, m_mean.detach(), t_lv, m_lv.detach())
                    rkl = rkl + cw * gaussian_analytical_kl(m_mean.detach(), mean, m_lv.detach(), lv)
                else:
                    last_sample = t_mean + a2
                output = torch.nan_to_num(layer(output) + last_sample)
        return output, rkl
