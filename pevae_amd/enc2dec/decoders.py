"""Attention / DecoderRNN with the reference's signatures (models/enc2dec/decoders.py:15-337).

TEACH_FORCE returns a `DeferredLogProbs`: a lazy stand-in for the reference's dense [B,L,V] log-prob tensor
(decoders.py:193,292 -- 1.02 GB fp32 at the Yelp shape).  `NLLEntropy` / `Perplexity` recognise it and route
to the fused MFMA GEMM + online-softmax + NLL kernel that never builds the logits; any other use
(`.materialize()`, indexing, torch functions) builds the dense tensor with the same GEMM core, so other callers
still see reference behaviour.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .base_modules import BaseRNN

TEACH_FORCE = "teacher_forcing"
TEACH_GEN = "teacher_gen"
GEN = "gen"
SUMS_ONLY = os.environ.get("PEVAE_VOCAB_SUMS_ONLY", "1") == "1"     # DeferredLogProbs._sums
SAMPLE_SITE = 0x5A000000        # Philox call sites of the fused sampling step (dropout sites count up from 0)


class Attention(nn.Module):
    """decoders.py:15-118.  forward(output [B,Lo,dec], context [B,Li,attn]) -> (output', attn [B,Lo,Li])."""

    def __init__(self, dec_size, attn_size, mode, project=False):
        super().__init__()
        self.mask = None
        self.mode = mode
        self.attn_size = attn_size
        self.dec_size = dec_size
        self.linear_out = nn.Linear(dec_size + attn_size, dec_size) if project else None
        if mode == 'general':
            self.attn_w = nn.Linear(dec_size, attn_size)
        elif mode == 'cat':
            self.dec_w = nn.Linear(dec_size, dec_size)
            self.attn_w = nn.Linear(attn_size, dec_size)
            self.query_w = nn.Linear(dec_size, 1)

    def forward(self, output, context):
        if context.size(0) != output.size(0):
            context = context.repeat(output.size(0) // context.size(0), 1, 1)
        batch_size, input_size = output.size(0), context.size(1)
        if self.mode == 'dot':
            attn = ops.bmm(output, context.transpose(1, 2))
        elif self.mode == 'general':
            attn = ops.bmm(ops.linear(output, self.attn_w.weight, self.attn_w.bias), context.transpose(1, 2))
        elif self.mode == 'cat':
            attn = ops.additive_scores(ops.linear(output, self.dec_w.weight, self.dec_w.bias),
                                       ops.linear(context, self.attn_w.weight, self.attn_w.bias),
                                       self.query_w.weight.view(-1), self.query_w.bias)
        else:
            raise ValueError("Unknown attention")
        if self.mask is not None:
            attn = attn.masked_fill(self.mask, -float('inf'))
        attn = ops.softmax_lastdim(attn)
        mix = ops.bmm(attn, context)
        combined = torch.cat((mix, output), dim=2)
        if self.linear_out is None:
            return combined, attn
        out = ops.tanh(ops.linear(combined.view(-1, self.dec_size + self.attn_size), self.linear_out.weight,
                                  self.linear_out.bias))
        return out.view(batch_size, -1, self.dec_size), attn


class DeferredLogProbs(object):
    """Lazy [B,L,V] log-softmax of (hid W^T + b) / temperature."""

    def __init__(self, hid, weight, bias, inv_temp, hid_bf16=None, layout=None, drop=None):
        # drop = (p, site, mask): the decoder's output dropout (decoders.py:189) has NOT been applied to `hid` yet; the
        # fused path folds it into the bf16 operand pack, the dense path applies it first
        self.hid, self.weight, self.bias, self.inv_temp = hid, weight, bias, inv_temp
        self.hid_bf16, self.layout, self.drop = hid_bf16, layout, drop
        self._dense = None
        self._nll_cache = {}

    # -- what the criterions use -------------------------------------------------------------------------
    def token_nll(self, labels, ignore_index, sums_only=False):
        """(nll_tok, targets in row order): per-token -log p[label] (0 where label == ignore_index) without the logits.
        sums_only: the caller only wants {sum, live count}; where the fused kernels run, the first result is then that pair
        already (third result True) and the per-token vector is never put back into the caller's row order."""
        key = (labels.data_ptr(), labels._version, tuple(labels.shape), ignore_index)
        hit = self._nll_cache.get(key)
        if hit is not None:
            return hit + (False,)
        B, L, H = self.hid.shape
        if self.hid_bf16 is not None or (self.drop is not None and self.layout is not None):
            Lh, Bh, Bpad = self.layout
            pre = ops.take_precompact(labels, tuple(self.layout), ignore_index)      # made at the start of the forward pass
            if pre is not None:
                tgt, pre = pre[0], (pre[1], pre[2])
            else:
                tgt = labels.reshape(B, L).t()
                if Bpad != B:
                    pad = torch.full((L, Bpad - B), ignore_index, dtype=labels.dtype, device=labels.device)
                    tgt = torch.cat([tgt, pad], dim=1)
                tgt = tgt.contiguous().view(-1)
            nll, _ = ops.VocabNLL.apply(self.hid, self.weight, self.bias, tgt, self.inv_temp, ignore_index,
                                        self.hid_bf16, self.layout, self.drop, pre, sums_only)
        elif self.drop is not None:
            tgt = labels.reshape(-1)
            nll, _ = ops.VocabNLL.apply(self.hid, self.weight, self.bias, tgt, self.inv_temp, ignore_index, None, None,
                                        self.drop, None, sums_only)
        else:
            tgt = labels.reshape(-1)
            nll, _ = ops.VocabNLL.apply(self.hid.reshape(B * L, H), self.weight, self.bias, tgt, self.inv_temp,
                                        ignore_index, None, None, None, None, sums_only)
        if sums_only and ops.VocabNLL.RETURNED_SUMS:     # the op says which form it returned (T = 2 rows look the same)
            return nll, tgt, True
        res = (nll, tgt)
        self._nll_cache[key] = res
        return res + (False,)

    def _sums(self, labels, ignore_index):
        key = ("sums", labels.data_ptr(), labels._version, tuple(labels.shape), ignore_index)
        hit = self._nll_cache.get(key)
        if hit is None:
            out, tgt, is_sums = self.token_nll(labels, ignore_index, sums_only=SUMS_ONLY)
            hit = self._nll_cache[key] = out if is_sums else ops.MaskedSum.apply(out, tgt, ignore_index)
        return hit

    def token_nll_sums(self, labels, ignore_index):
        """(sum of the token NLL, number of non-ignored tokens) as device scalars; NLLEntropy and Perplexity of one step
        ask for the same pair, which is reduced once."""
        hit = self._sums(labels, ignore_index)
        return hit[0], hit[1]

    def loss_head(self, labels, ignore_index):
        """(NLLEntropy avg_type 'seq', Perplexity) of these labels as device scalars, one launch for both (ops.NllHead);
        the two criterions of a step ask for one each."""
        key = ("head", labels.data_ptr(), labels._version, tuple(labels.shape), ignore_index)
        hit = self._nll_cache.get(key)
        if hit is None:
            hit = self._nll_cache[key] = ops.NllHead.apply(self._sums(labels, ignore_index), self.hid.shape[0])
        return hit

    # -- dense fallback: behaves like the reference's tensor ---------------------------------------------------
    def materialize(self):
        if self._dense is None:
            B, L, H = self.hid.shape
            hid = self.hid if self.drop is None else ops.DropoutFn.apply(self.hid, *self.drop)
            flat = ops.VocabLogProbDense.apply(hid.reshape(B * L, H), self.weight, self.bias, self.inv_temp)
            self._dense = flat.view(B, L, -1)
        return self._dense

    def size(self, dim=None):
        shape = torch.Size((self.hid.shape[0], self.hid.shape[1], self.weight.shape[0]))
        return shape if dim is None else shape[dim]

    @property
    def shape(self):
        return self.size()

    def dim(self):
        return 3

    def __getattr__(self, name):            # anything else: act as the dense tensor
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __getitem__(self, idx):
        return self.materialize()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        conv = lambda a: a.materialize() if isinstance(a, DeferredLogProbs) else a
        args = tuple(conv(a) if not isinstance(a, (list, tuple)) else type(a)(conv(x) for x in a) for a in args)
        kwargs = {k: conv(v) for k, v in kwargs.items()}
        return func(*args, **kwargs)


def _repeat_layers(state, n_layers):
    """init_state.repeat(n_layers, 1, 1) (decoders.py:226-232); for one layer that is the tensor itself -- no copy forward,
    no sum backward."""
    return state if n_layers == 1 else state.repeat(n_layers, 1, 1)


class DecoderRNN(BaseRNN):
    """decoders.py:121-337.  Deviations from the reference, both needed to make documented features usable:
    (1) `Attention` IS built when use_attention=True (the reference forces it off at decoders.py:154 yet keeps
    self.use_attention, so passing attn_context raises there); (2) `unk_mask` lives on the parameters' device
    instead of an unconditional `.cuda()` (decoders.py:144)."""

    def __init__(self, vocab_size, max_len, input_size, hidden_size, sos_id, eos_id, unk_id, n_layers=1,
                 rnn_cell='lstm', input_dropout_p=0, dropout_p=0, use_attention=False, attn_mode='cat',
                 attn_size=None, use_gpu=True, embedding=None, output_size=None, tie_output_embed=False,
                 init_range=0.1, mask_unk=False, softmax_temperature=1.0):
        super().__init__(vocab_size, input_size, hidden_size, input_dropout_p, dropout_p, n_layers, rnn_cell, False)
        self.output_size = vocab_size if output_size is None else output_size
        self.max_length = max_len
        self.use_attention = use_attention
        self.eos_id = eos_id
        self.sos_id = sos_id
        self.init_input = None
        self.use_gpu = use_gpu
        self.unk_id = unk_id
        self.softmax_temperature = softmax_temperature
        onehot = torch.zeros(self.output_size)
        onehot[self.unk_id] = -np.inf
        self.register_buffer("unk_mask", onehot, persistent=False)
        self.mask_unk = mask_unk
        self.embedding = nn.Embedding(vocab_size, self.input_size) if embedding is None else embedding
        if use_attention:
            self.attention = Attention(self.hidden_size, attn_size, attn_mode, project=True)
        self.tie_output_embed = tie_output_embed
        if not tie_output_embed:
            self.project = nn.Linear(self.hidden_size, self.output_size)
        self.output_dropout = ops.Dropout(p=input_dropout_p)           # decoders.py:166
        for para in self.parameters():                                   # decoders.py:167-168 (incl. shared embedding)
            nn.init.uniform_(para.data, -init_range, init_range)

    def _out_weights(self):
        if self.tie_output_embed:                                        # decoders.py:159-160
            w = self.embedding.weight
            return w, torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        return self.project.weight, self.project.bias

    def _hidden_states(self, input_var, hidden, encoder_outputs, latent_variable, defer_output_dropout=False):
        """decoders.py:171-189: embedding -> dropout -> [cat latent] -> RNN -> [attention] -> dropout.
        defer_output_dropout: return the states BEFORE the output dropout plus the (p, site, mask) it must still get (the
        teacher-forced path folds it into the vocabulary operand pack); the 4th result is then (hseq, drop)."""
        # the teacher-forced input embedding on the side stream: its backward (a sort and a segmented sum nobody waits
        # for) then runs beside the latent stack's backward instead of in front of it
        side = latent_variable is None and input_var.size(1) > 1 and torch.is_grad_enabled() and input_var.is_cuda
        leaf = None
        ahead = self.__dict__.pop("_input_ahead", None)
        if ahead is not None and ahead[0] == (input_var.data_ptr(), input_var._version, tuple(input_var.shape)) and side:
            leaf = ahead[1]                 # prepare_teacher_forced() made it beside the latent stack
        elif (side and ops.LEAF_DECODER_INPUT and self.rnn.cell == "gru" and self.rnn.num_layers == 1
                and not self.rnn.bidirectional):
            # with a gradient sink on the table: no autograd nodes for the input at all, its gradient is taken to the table
            # inside the recurrence's postponed side-stream work
            leaf = ops.embed_dropout_leaf(self.embedding.weight, input_var, self.embedding.padding_idx, self.input_dropout)
        if leaf is not None:
            embedded, consumer = leaf
            output, hidden, hseq = self.rnn(embedded, hidden, dx_consumer=consumer)
        else:
            with ops.on_side(side):
                embedded = ops.embedding(self.embedding.weight, input_var, self.embedding.padding_idx)
                embedded = self.input_dropout(embedded)
            if latent_variable is not None:
                tiled = latent_variable.unsqueeze(1).repeat(1, embedded.size(1), 1)
                embedded = torch.cat([embedded, tiled], dim=-1)
            output, hidden, hseq = self.rnn(embedded, hidden, x_side=side)
        attn = None
        if self.use_attention and encoder_outputs is not None:
            output, attn = self.attention(output, encoder_outputs)
            hseq = None
        dropping = self.training and self.output_dropout.p > 0
        if defer_output_dropout:
            drop = (self.output_dropout.p, self.output_dropout.next_site(), self.output_dropout.mask) if dropping else None
            return output, hidden, attn, (None if dropping else hseq, drop)
        if dropping:
            hseq = None
        output = self.output_dropout(output)
        return output, hidden, attn, hseq

    def prepare_teacher_forced(self, input_var):
        """Ahead of forward(), on whatever stream the caller is on (DLSVAE._prepare_on_side: the side stream, beside the
        latent stack): the teacher-forced input dropout(embedding(ids)) as a leaf value and the recurrence's input projection
        of it -- neither depends on the latent.  _hidden_states() / GRULayer.forward pick them up once; nothing happens when
        the leaf path does not apply (no gradient sink on the table, LSTM, stacked or attention-fed inputs)."""
        self.__dict__.pop("_input_ahead", None)
        if not (ops.LEAF_DECODER_INPUT and ops.HOIST_DECODER_INPUT and self.rnn.cell == "gru" and self.rnn.num_layers == 1
                and not self.rnn.bidirectional and input_var.is_cuda and input_var.size(1) > 1 and torch.is_grad_enabled()):
            return
        leaf = ops.embed_dropout_leaf(self.embedding.weight, input_var, self.embedding.padding_idx, self.input_dropout)
        if leaf is None:
            return
        w_ih, _w_hh, b_ih, _b_hh = self.rnn.layer_weights(0, 0)
        ops.preproject_gru_input(leaf[0], w_ih, b_ih, self.rnn.hidden_size)
        self.__dict__["_input_ahead"] = ((input_var.data_ptr(), input_var._version, tuple(input_var.shape)), leaf)

    def forward_step(self, input_var, hidden, encoder_outputs, latent_variable=None):
        """Dense variant with the reference's return value (log-probs [B,Lout,V], hidden, attn)."""
        output, hidden, attn, _ = self._hidden_states(input_var, hidden, encoder_outputs, latent_variable)
        w, b = self._out_weights()
        B, Lo, H = output.shape
        logp = ops.VocabLogProbDense.apply(output.reshape(B * Lo, H), w, b, 1.0 / self.softmax_temperature)
        return logp.view(B, Lo, -1), hidden, attn

    def forward(self, batch_size, inputs=None, init_state=None, attn_context=None, mode=TEACH_FORCE,
                gen_type='greedy', beam_size=4, latent_variable=None, feature_input=None):
        if (mode == GEN and (gen_type == 'greedy' or (gen_type == 'sample' and getattr(self, "sample_noise", None) is None))
                and getattr(self, "persistent_decode", False) and not self.training
                and not torch.is_grad_enabled() and attn_context is None and not self.use_attention
                and torch.is_tensor(init_state) and self.rnn.cell == "gru" and self.rnn.num_layers == 1
                and not self.rnn.bidirectional):
            return self._persistent_greedy(batch_size, inputs, init_state, latent_variable, gen_type)
        if (mode == GEN and (gen_type == 'greedy' or (gen_type == 'sample' and getattr(self, "sample_noise", None) is None)
                             or (gen_type == 'beam' and beam_size <= ops.VOCAB_TOPK_MAX))
                and getattr(self, "graph_decode", False)
                and getattr(self, "fused_greedy", False) and not self.training and not torch.is_grad_enabled()
                and attn_context is None and not self.use_attention and torch.is_tensor(init_state)):
            return self._graphed_greedy(batch_size, inputs, init_state, latent_variable, gen_type,
                                        beam_size if gen_type == 'beam' else 1)
        ret_dict = dict()
        self.max_length = inputs.size(1)                # decoders.py:201 (GEN still needs `inputs` for this)
        if self.use_attention:
            ret_dict[DecoderRNN.KEY_ATTN_SCORE] = list()
        if mode == GEN:
            inputs = None
        if gen_type != 'beam':
            beam_size = 1
        dev = self.embedding.weight.device
        if inputs is not None:
            decoder_input = inputs
        else:
            decoder_input = torch.full((batch_size * beam_size, 1), self.sos_id, dtype=torch.long, device=dev)
        lstm = self.rnn_cell is nn.LSTM
        if mode == GEN and gen_type == 'beam':
            if lstm:
                decoder_hidden = tuple(self.repeat_state(s, batch_size, beam_size) for s in init_state)
            else:
                decoder_hidden = self.repeat_state(init_state, batch_size, beam_size)
        elif init_state is None:
            decoder_hidden = None
        elif lstm:
            decoder_hidden = tuple(_repeat_layers(s, self.n_layers) for s in init_state)
        else:
            decoder_hidden = _repeat_layers(init_state, self.n_layers)

        sequence_symbols, back_pointers = [], []
        lengths = np.array([self.max_length] * batch_size * beam_size)

        if mode == TEACH_FORCE:
            output, decoder_hidden, attn, (hseq, drop) = self._hidden_states(decoder_input, decoder_hidden, attn_context,
                                                                             latent_variable, defer_output_dropout=True)
            w, b = self._out_weights()
            B, L, H = output.shape
            hid_bf16 = layout = None
            if hseq is not None:
                Hk, K3, Bpad, NG = ops.gru_geometry(B, H)
                hid_bf16, layout = hseq[Bpad:], (L, B, Bpad)
            elif drop is not None and not (self.use_attention and attn_context is not None):
                # the states are a strided view of the recurrence's time-major buffer: pack them time-major as well
                layout = (L, B, ops.gru_geometry(B, H)[2])
            decoder_outputs = DeferredLogProbs(output, w, b, 1.0 / self.softmax_temperature, hid_bf16, layout, drop)
        else:
            step_outputs, cum_sum = [], None
            fused = getattr(self, "fused_greedy", False) and gen_type in ('greedy', 'sample')
            fused_beam = getattr(self, "fused_greedy", False) and gen_type == 'beam' and beam_size <= ops.VOCAB_TOPK_MAX
            if fused and gen_type == 'sample' and getattr(self, "sample_noise", None) is None:
                ops.rng_advance(dev)                 # a new decode draws new symbols; token di is call site SAMPLE_SITE + di
            # the projection / recurrence weights are packed to bf16 once for the whole decode, not once per token
            with ops.packed_weights():
                for di in range(self.max_length):
                    if fused:
                        # opt-in (self.fused_greedy = True): arg-max (or the multinomial draw of gen_type 'sample') fused into
                        # the projection GEMM, no [B,V] log-probs; the
                        # returned decoder_outputs is then [B, L] = log-probability of each emitted symbol (a deviation from
                        # the reference's [B, L, V] return value, for callers that only consume ret_dict['sequence'])
                        output, decoder_hidden, step_attn, _ = self._hidden_states(decoder_input, decoder_hidden,
                                                                                   attn_context, latent_variable)
                        w, b = self._out_weights()
                        if gen_type == 'sample':     # torch.multinomial as arg-max of p / q, q ~ Exp(1), fused the same way
                            noise = getattr(self, "sample_noise", None)
                            symbols, lp = ops.vocab_sample(output.reshape(-1, output.shape[-1]), w, b,
                                                           1.0 / self.softmax_temperature,
                                                           self.unk_id if self.mask_unk else -1,
                                                           noise=None if noise is None else noise[di], site=SAMPLE_SITE + di)
                        else:
                            symbols, lp = ops.vocab_argmax(output.reshape(-1, output.shape[-1]), w, b,
                                                           1.0 / self.softmax_temperature,
                                                           self.unk_id if self.mask_unk else -1)
                        symbols = symbols.view(-1, 1)
                        step_outputs.append(lp.view(-1, 1))
                        if self.use_attention:
                            ret_dict[DecoderRNN.KEY_ATTN_SCORE].append(step_attn)
                        sequence_symbols.append(symbols)
                        decoder_input = symbols
                        continue
                    if fused_beam:
                        # opt-in (self.fused_greedy = True): the beam step without [B * beam, V] rows.  Per row the `beam` best
                        # symbols and their log-probabilities come out of the projection GEMM's slab epilogue
                        # (ops.vocab_topk); the beam * beam candidates of a batch element are ranked here.  Everything else
                        # as below (:263-270): symbols / back pointers per step, states NOT re-ordered, as in the reference.
                        output, decoder_hidden, step_attn, _ = self._hidden_states(decoder_input, decoder_hidden,
                                                                                   attn_context, latent_variable)
                        w, b = self._out_weights()
                        k = beam_size
                        ids_k, lp_k = ops.vocab_topk(output.reshape(-1, output.shape[-1]), w, b, k,
                                                     1.0 / self.softmax_temperature, self.unk_id if self.mask_unk else -1)
                        if di == 0:              # all beams hold the same state: the first beam's row (:264 [:, 0:V])
                            top_v = lp_k.view(batch_size, beam_size, k)[:, 0]
                            symbols = ids_k.view(batch_size, beam_size, k)[:, 0].reshape(-1, 1)
                            back = torch.zeros((batch_size * beam_size, 1), dtype=torch.long, device=dev)
                            emitted = top_v
                        else:
                            cand = (cum_sum.view(batch_size, beam_size, 1) + lp_k.view(batch_size, beam_size, k))
                            top_v, pos = cand.view(batch_size, beam_size * k).topk(beam_size)
                            back = torch.div(pos, k, rounding_mode='floor').view(-1, 1)
                            symbols = ids_k.view(batch_size, beam_size * k).gather(1, pos).view(-1, 1)
                            emitted = lp_k.view(batch_size, beam_size * k).gather(1, pos)
                        back_pointers.append(back)
                        cum_sum = top_v.reshape(-1, 1)
                        step_outputs.append(emitted.reshape(-1, 1))
                        if self.use_attention:
                            ret_dict[DecoderRNN.KEY_ATTN_SCORE].append(step_attn)
                        sequence_symbols.append(symbols)
                        decoder_input = symbols
                        continue
                    step_output, decoder_hidden, step_attn = self.forward_step(decoder_input, decoder_hidden,
                                                                               attn_context, latent_variable)
                    step_outputs.append(step_output)
                    scores = step_output.squeeze(1)
                    if self.mask_unk:
                        scores = scores + self.unk_mask
                    if self.use_attention:
                        ret_dict[DecoderRNN.KEY_ATTN_SCORE].append(step_attn)
                    if gen_type == 'greedy':
                        symbols = scores.topk(1)[1]
                    elif gen_type == 'sample':
                        # base_modules.py:54-59 torch.multinomial(probs, 1) IS argmax(probs / q), q ~ Exp(1) per entry (the
                        # sampler torch uses for one draw per row); `sample_noise` ([L, B, V], set by parity tests)
                        # supplies q so that the draw can be compared with the reference's
                        noise = getattr(self, "sample_noise", None)
                        q = noise[di] if noise is not None else torch.empty_like(scores).exponential_()
                        symbols = (torch.exp(scores) / q).argmax(-1, keepdim=True)
                    elif gen_type == 'beam':
                        if di == 0:
                            seq_score = scores.view(batch_size, -1)[:, 0:self.output_size]
                        else:
                            seq_score = (cum_sum + scores).view(batch_size, -1)
                        top_v, top_id = seq_score.topk(beam_size)
                        back_pointers.append(torch.div(top_id, self.output_size, rounding_mode='floor').view(-1, 1))
                        symbols = top_id.fmod(self.output_size).view(-1, 1)
                        cum_sum = top_v.view(-1, 1)
                    else:
                        raise ValueError("Unsupported decoding mode")
                    sequence_symbols.append(symbols)
                    decoder_input = symbols
            decoder_outputs = torch.cat(step_outputs, dim=1)
            # lengths (decoders.py:256-259 updates them with one device->host copy per token): position of the first EOS
            # + 1, max_length without one -- formed on the device after the loop, one copy per decode
            before_eos = (torch.cat(sequence_symbols, dim=1).eq(self.eos_id).cumsum(1) == 0).sum(1)
            lengths = torch.clamp(before_eos + 1, max=self.max_length)
            if not getattr(self, "_device_lengths", False):
                lengths = lengths.cpu().numpy()
            if gen_type == 'beam':
                best = cum_sum.view(-1, beam_size).topk(1)[1].view(-1)
                rows = torch.arange(batch_size, device=dev)
                final = []
                for symbols, back_ptrs in zip(sequence_symbols[::-1], back_pointers[::-1]):
                    final.append(symbols.view(-1, beam_size)[rows, best].unsqueeze(1))
                    best = back_ptrs.view(-1, beam_size)[rows, best]
                sequence_symbols = final[::-1]

        ret_dict[DecoderRNN.KEY_SEQUENCE] = sequence_symbols
        ret_dict[DecoderRNN.KEY_LENGTH] = lengths if getattr(self, "_device_lengths", False) else lengths.tolist()
        return decoder_outputs, decoder_hidden, ret_dict

    def _graphed_greedy(self, batch_size, inputs, init_state, latent_variable, gen_type='greedy', beam_size=1):
        """Opt-in (self.graph_decode = self.fused_greedy = True, eval mode, no grad, GRU cell, no attention): the whole
        free-running greedy (or `sample`, or fused `beam`) decode -- max_length x (embedding, input projection, one recurrence
        step, projection + arg-max / multinomial draw / top-k and candidate ranking; for beam also the back-tracking) --
        captured once per (mode, batch, length) into a HIP graph and replayed; the only
        host work per decode is the copy of the initial state in and one read-back of the lengths.  Weights are re-packed
        inside the graph, so a replay follows parameter updates; the sampler's step counter lives in device memory and
        is advanced inside the graph, so every replay of a `sample` decode draws anew.  Same return value as the fused
        per-token path."""
        L = inputs.size(1)
        key = (gen_type, beam_size, batch_size, L, tuple(init_state.shape),
               None if latent_variable is None else tuple(latent_variable.shape))
        cache = self.__dict__.setdefault("_decode_graphs", {})
        ent = cache.get(key)
        if ent is None:
            st = init_state.detach().clone()
            lt = None if latent_variable is None else latent_variable.detach().clone()
            width = torch.zeros((batch_size, L), dtype=torch.long, device=st.device)     # only its width is read
            self.graph_decode, self._device_lengths = False, True
            try:
                self.forward(batch_size, width, st, None, GEN, gen_type, beam_size, lt)     # warm-up: every workspace exists
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    outs, hid, ret = self.forward(batch_size, width, st, None, GEN, gen_type, beam_size, lt)
                    seq = torch.cat(ret[DecoderRNN.KEY_SEQUENCE], dim=1)
                    lens = ret[DecoderRNN.KEY_LENGTH]
            finally:
                self.graph_decode, self._device_lengths = True, False
            ent = cache[key] = (graph, st, lt, outs, hid, seq, lens)
        graph, st, lt, outs, hid, seq, lens = ent
        st.copy_(init_state)
        if lt is not None:
            lt.copy_(latent_variable)
        graph.replay()
        seq_out = seq.clone()
        ret_dict = {DecoderRNN.KEY_SEQUENCE: list(seq_out.split(1, dim=1)), DecoderRNN.KEY_LENGTH: lens.cpu().tolist()}
        return outs.clone(), hid.clone(), ret_dict

    def _persistent_greedy(self, batch_size, inputs, init_state, latent_variable, gen_type='greedy'):
        """Opt-in (self.persistent_decode = True; eval mode, no grad, one-layer GRU, no attention): the whole free-running
        greedy (or `sample`) decode as ONE persistent kernel launch (ops.greedy_decode / pv_greedy_decode / pv_sample_decode,
        SURVEY section 8 row f3) -- no per-token launches at all, W_hh and W_proj resident in LDS across tokens.  Same return
        value as the fused per-token path: decoder_outputs [B, L] = log-probability of each emitted symbol; a `sample`
        decode advances the device generator first and draws token t at call site SAMPLE_SITE + t, exactly as the per-token
        fused path does."""
        L = inputs.size(1)
        sample = gen_type == 'sample'
        if sample:
            ops.rng_advance(self.embedding.weight.device)
        self.max_length = L
        w_ih, w_hh, b_ih, b_hh = self.rnn.layer_weights(0, 0)
        w, b = self._out_weights()
        h0 = init_state.reshape(-1, init_state.shape[-1])[:batch_size]
        seq, lp, hn = ops.greedy_decode(self.embedding.weight, w_ih, w_hh, b_ih, b_hh, w, b, h0, L, self.sos_id,
                                        latent=latent_variable, inv_temp=1.0 / self.softmax_temperature,
                                        mask_col=self.unk_id if self.mask_unk else -1, sample=sample, site0=SAMPLE_SITE)
        before_eos = (seq.eq(self.eos_id).cumsum(1) == 0).sum(1)
        lengths = torch.clamp(before_eos + 1, max=L)
        ret_dict = {DecoderRNN.KEY_SEQUENCE: list(seq.split(1, dim=1)),
                    DecoderRNN.KEY_LENGTH: lengths if getattr(self, "_device_lengths", False) else lengths.cpu().tolist()}
        return lp, hn.unsqueeze(0), ret_dict
