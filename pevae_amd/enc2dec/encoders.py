"""EncoderRNN / RnnUttEncoder with the reference's signatures (models/enc2dec/encoders.py:7-135) on the
persistent HIP GRU kernel."""
import torch
import torch.nn as nn

from .. import ops
from .base_modules import BaseRNN


class EncoderRNN(BaseRNN):
    """forward(input_var [B,L,in], input_lengths=None, init_state=None) -> (output [B,L,H*dirs],
    hidden [layers*dirs,B,H])  -- encoders.py:43-66."""

    def __init__(self, input_size, hidden_size, input_dropout_p=0, dropout_p=0, n_layers=1, rnn_cell='gru',
                 variable_lengths=False, bidirection=False):
        super().__init__(-1, input_size, hidden_size, input_dropout_p, dropout_p, n_layers, rnn_cell, bidirection)
        self.variable_lengths = variable_lengths
        self.output_size = hidden_size * 2 if bidirection else hidden_size

    def forward(self, input_var, input_lengths=None, init_state=None):
        embedded = self.input_dropout(input_var)
        # encoders.py:55-65: pack_padded_sequence -> rnn -> pad_packed_sequence; the HIP recurrence freezes each row at
        # its length instead of packing
        lengths = input_lengths if self.variable_lengths else None
        output, hidden, _ = self.rnn(embedded, init_state, lengths)
        return output, hidden


class RnnUttEncoder(nn.Module):
    """encoders.py:69-135: embedding + EncoderRNN + optional additive self-attention pooling.  (Imported by
    the reference's dialog_models.py:8 but never instantiated there.)"""

    def __init__(self, utt_cell_size, dropout, rnn_cell='gru', bidirection=True, use_attn=False, embedding=None,
                 vocab_size=None, embed_dim=None, feat_size=0):
        super().__init__()
        self.bidirection = bidirection
        self.utt_cell_size = utt_cell_size
        if embedding is None:
            self.embed_size = embed_dim
            self.embedding = nn.Embedding(vocab_size, embed_dim)
        else:
            self.embedding = embedding
            self.embed_size = embedding.embedding_dim
        self.rnn = EncoderRNN(self.embed_size + feat_size, utt_cell_size, 0.0, dropout, rnn_cell=rnn_cell,
                              variable_lengths=False, bidirection=bidirection)
        self.multipler = 2 if bidirection else 1
        self.output_size = self.utt_cell_size * self.multipler
        self.use_attn = use_attn
        self.feat_size = feat_size
        if use_attn:
            self.key_w = nn.Linear(self.output_size, self.utt_cell_size)
            self.query = nn.Linear(self.utt_cell_size, 1)

    def forward(self, utterances, feats=None, init_state=None, return_all=False):
        B, n_ctx, n_tok = (int(s) for s in utterances.size())
        if init_state is not None:
            init_state = init_state.repeat(1, n_ctx, 1)
        words = ops.embedding(self.embedding.weight, utterances.view(-1, n_tok), self.embedding.padding_idx)
        if feats is not None:
            words = torch.cat([words, feats.view(-1, 1).unsqueeze(1).repeat(1, n_tok, 1)], dim=2)
        enc_outs, enc_last = self.rnn(words, init_state=init_state)
        if self.use_attn:
            fc1 = torch.tanh(ops.linear(enc_outs, self.key_w.weight, self.key_w.bias))
            score = ops.linear(fc1, self.query.weight, self.query.bias).squeeze(2)
            attn = torch.softmax(score, score.dim() - 1).unsqueeze(2)
            utt = torch.sum(attn * enc_outs, dim=1)
        else:
            attn = None
            utt = enc_last.transpose(0, 1).contiguous().view(-1, self.output_size)
        utt = utt.view(B, n_ctx, self.output_size)
        if return_all:
            return utt, enc_outs, enc_last, attn
        return utt
