"""The composed DLS_VAE training step on the HIP modules.

The reference's `DLS_VAE` class cannot be constructed (it needs network downloads and undefined names,
dialog_models.py:3356-3437), so this is the build's documented composition:
  * data flow of FS_VAE.forward (dialog_models.py:755-818): user / item / rating / word embedding gathers
    (:758-765), EncoderRNN over the explanation (:765-766), recognition and prior heads, reparameterised
    sample (:191-203), LinearConnector into the decoder state (:711, nn_lib.py:146-156), teacher-forced
    DecoderRNN sharing the word embedding (:712, :560-570), NLLEntropy + Perplexity (:790-791);
  * loss / Pack contract of DLS_VAE (dialog_models.py:3451-3465, :3569): Pack(vae_zkl, vae_nll, vae_PPL,
    vae_rnll, vae_rkl); valid_loss = vae_nll + vae_rkl in training, vae_nll in eval.
Latent stack 'gauss' (flat): q(z|x) = Hidden2Gaussian(h_enc), p(z|u,i,r) = Hidden2Gaussian(tanh(W [u;i;r])),
KL = NormKLLoss(q, p).  Latent stack 'dls' (config.latent_stack = "dls"): the reference's DLS pyramid, wired as
DLS_VAE.__init__ / forward do (dialog_models.py:3401-3417, :3529-3538): DlsRec / DlsPri over the response / condition
states (H -> H/16, activations kept), DLS_Auto latent fusion over the three deepest levels -> z [B, H/4] and rkl, then
x_init_connector = Block(H/4, H/2) -> Block(H/2, H) into the decoder state; rkl_weight fixed at 1 (the reference's
adaptive rkl_weight, :3564-3567, uses attributes that are never defined).  oracle/ref_step.py holds the same composition on the CPU oracle modules.
"""
import os
from argparse import Namespace

import torch
import torch.nn as nn

from . import criterions, dls_latent as L, nn_lib, ops
from .enc2dec.decoders import GEN, TEACH_FORCE, DecoderRNN
from .enc2dec.encoders import EncoderRNN
from .utils import Pack

PREPARE_LATE = os.environ.get("PEVAE_PREPARE_LATE", "1") == "1"
# the user / item / rating lookups (and the spectral pre-pass) beside the encoder's forward recurrence instead of in front of it
HEAD_SIDE = os.environ.get("PEVAE_HEAD_SIDE", "1") == "1"
PREPARE_ON_SIDE = os.environ.get("PEVAE_PREPARE_ON_SIDE", "1") == "1"      # DLSVAE._prepare_on_side

# vocab = [<pad>, <unk>, <sys>, <usr>] + words by frequency (corpora.py:649).  <s> and </s> open and close every
# sentence (corpora.py:620), so on a real corpus they are the two most frequent words: ids 4 and 5.  The model takes
# them from the vocabulary, as FS_VAE.__init__ does (dialog_models.py:455-456: go_id = rev_vocab[BOS], eos_id =
# rev_vocab[EOS]); these are the defaults for synthetic data, which has no vocabulary.
PAD_ID, UNK_ID, SYS_ID, USR_ID, BOS_ID, EOS_ID = 0, 1, 2, 3, 4, 5
N_SPECIAL = 6                                    # synthetic word ids start here


def default_config(**kw):
    cfg = Namespace(vocab_size=20004, embed_size=100, dec_cell_size=512, max_seq_len=50, n_user=30000,
                    n_item=20000, latent_size=64, dropout=0.3, rnn_cell="gru", num_layer=1, use_attention=False,
                    attn_type="cat", do_pred=False, beam_size=10, batch_size=256, init_lr=1e-4, grad_clip=5.0,
                    warmup_steps=0, use_gpu=True, latent_stack="gauss", expand_weight=0.5, converge_weight=0.1,
                    cip_weight=1.0, ost=False, pad_id=PAD_ID, unk_id=UNK_ID, bos_id=BOS_ID, eos_id=EOS_ID,
                    use_mutual=False, use_feature=False, n_feature=0, den_weight=1.0,
                    global_negatives=False)     # batch sharding only: batch-coupled losses over gathered rows (dp.GatherRows)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def config_from_vocab(config, rev_vocab):
    """Special-symbol ids and vocabulary size from a corpus vocabulary (dialog_models.py:451-457: vocab_size =
    len(vocab), go_id = rev_vocab[BOS], eos_id = rev_vocab[EOS], unk via corpora.py:651)."""
    from .dataset.data_loaders import BOS, EOS, PAD, UNK
    config.vocab_size = len(rev_vocab)
    config.pad_id, config.unk_id = rev_vocab[PAD], rev_vocab[UNK]
    config.bos_id, config.eos_id = rev_vocab[BOS], rev_vocab[EOS]
    return config


class DLSVAE(nn.Module):
    def __init__(self, config):
        super().__init__()
        c = self.config = config
        H, E, Z = c.dec_cell_size, c.embed_size, c.latent_size
        self.emb_size = H // 2
        self.pad_id, self.unk_id = getattr(c, "pad_id", PAD_ID), getattr(c, "unk_id", UNK_ID)
        self.go_id, self.eos_id = getattr(c, "bos_id", BOS_ID), getattr(c, "eos_id", EOS_ID)      # :455-456
        self.x_embedding = nn.Embedding(c.vocab_size, E, padding_idx=self.pad_id)       # dialog_models.py:468
        self.user_embedding = nn.Embedding(c.n_user, self.emb_size)                      # :473
        self.item_embedding = nn.Embedding(c.n_item, self.emb_size)                      # :475
        self.rating_embedding = nn.Embedding(5, self.emb_size)                           # :502
        self.x_encoder = EncoderRNN(E, H, dropout_p=c.dropout, rnn_cell=c.rnn_cell, variable_lengths=False)  # :530
        self.use_feature = bool(getattr(c, "use_feature", False))
        if self.use_feature:                                                             # :488, :555-558, :762-763
            self.feature_embedding = nn.Embedding(c.n_feature, self.emb_size)
        self.x_decoder = DecoderRNN(c.vocab_size, c.max_seq_len, E + (self.emb_size if self.use_feature else 0), H,
                                    self.go_id, self.eos_id, self.unk_id,
                                    n_layers=c.num_layer, rnn_cell=c.rnn_cell, input_dropout_p=c.dropout,
                                    dropout_p=c.dropout, use_attention=c.use_attention, attn_mode=c.attn_type,
                                    attn_size=H, use_gpu=c.use_gpu, embedding=self.x_embedding)               # :560-570
        self.latent_stack = getattr(c, "latent_stack", "gauss")
        self.use_mutual = bool(getattr(c, "use_mutual", False)) and self.latent_stack == "dls"
        if self.latent_stack == "dls":                                                   # dialog_models.py:3401-3417
            self.rec_encoder = L.DlsRec(H, H // 16, c)
            self.pri_encoder = L.DlsPri(H, H // 16, c)
            if self.use_mutual:
                # --use_mutual (configs[4]): the DEM mutual-information stack, wired as DeN_VAE wires DeN_Auto
                # (dialog_models.py:2917-2931, :3089): two fusion levels with a discriminator each -> z [B, H/8], then
                # a three-Block connector H/8 -> H/4 -> H/2 -> H
                self.sup_decoder = L.DeNAuto(H // 32, H // 16, c)
                self.x_init_connector = nn.Sequential(L.Block(H // 8, H // 4, H // 4), L.Block(H // 4, H // 2, H // 2),
                                                      L.Block(H // 2, H, H))
            else:
                self.sup_decoder = L.DLSAuto(H // 32, H // 8, c)
                self.x_init_connector = nn.Sequential(L.Block(H // 4, H // 2, H // 2), L.Block(H // 2, H, H))
        self.prior_fc = nn.Linear(3 * self.emb_size, H)
        if self.latent_stack == "gauss":
            self.rec_head = nn_lib.Hidden2Gaussian(H, Z)
            self.pri_head = nn_lib.Hidden2Gaussian(H, Z)
            self.x_init_connector = nn_lib.LinearConnector(Z, H, c.rnn_cell == "lstm")  # :626
        if c.do_pred:                                                                    # `self.dnn`, used at :815, never
            self.dnn = nn.Sequential(nn.Linear(2 * self.emb_size, H), nn.Linear(H, 1))  # defined by the reference
        self.nll_loss = criterions.NLLEntropy(self.pad_id, c)                            # :634
        self.ppl = criterions.Perplexity(self.pad_id, c)                                 # :637
        self.kl_loss = criterions.NormKLLoss(unit_average=False)
        self.backLoss = True
        # split_backward: True / 1 = cut behind the vocabulary projection; 2 = also in front of the utterance encoder
        self.split_backward, self._cut, self._cut_enc = False, None, None

    # -- loss contract (dialog_models.py:3451-3465) ----------------------------------------------------------
    def valid_loss(self, loss, batch_cnt=None, step=None):
        loss.vae_zkl = 0
        if loss.vae_nll is None:
            loss.vae_nll = 0
        if self.training and loss.vae_rkl is not None:
            total = loss.vae_nll + loss.vae_rkl
        else:
            total = loss.vae_nll
        if self.config.do_pred and loss.get("rating_mse") is not None:
            total = total + loss.rating_mse
        return total

    def backward(self, batch_cnt, loss, step=None):                                     # dialog_models.py:51-53
        if self._cut is not None:
            self.backward_vocab(loss)
            return self.backward_rest(loss)
        # valid_loss() is a plain sum of its terms (:3451-3465): d total / d term = 1 for each, so the backward pass starts
        # from the terms themselves -- the sum is never formed (an add launch between the forward and the backward pass) --
        # with a cached scalar one as every seed (`.backward()` alone fills a fresh one-element tensor every step)
        terms = [loss.vae_nll]
        if self.training and loss.vae_rkl is not None:
            terms.append(loss.vae_rkl)
        if self.config.do_pred and loss.get("rating_mse") is not None:
            terms.append(loss.rating_mse)
        terms = [t for t in terms if torch.is_tensor(t) and t.requires_grad]
        if not terms:
            return
        seed = self.__dict__.get("_one")
        if seed is None or seed.device != terms[0].device or seed.dtype != terms[0].dtype:
            seed = self.__dict__["_one"] = torch.ones((), dtype=terms[0].dtype, device=terms[0].device)
        self._run_backward(lambda: torch.autograd.backward(terms, [seed] * len(terms)))

    @staticmethod
    def _run_backward(fn):
        """Backward pass with the recurrences' weight-gradient GEMMs off the critical path (ops.side_work); the side stream
        is joined before this returns, so the caller sees finished gradients on the current stream."""
        prev, ops.ASYNC_WGRAD = ops.ASYNC_WGRAD, True
        try:
            fn()
        finally:
            ops.ASYNC_WGRAD = prev
            ops.join_side_work()

    def backward_vocab(self, loss):
        """First part of the backward pass with split_backward: d(valid_loss)/d(vae_nll) = 1 (:3451-3465), so this is the
        vocabulary softmax / projection backward alone; it leaves dL/d(decoder states) in the cut leaf."""
        loss.vae_nll.backward()

    def backward_mid(self, loss):
        """Second part: decoder recurrence, latent stack, prior path.  With the encoder cut (split_backward = 2) it stops at
        the utterance encoder's output: every gradient but the encoder's and the word table's is final when it returns."""
        hid, leaf = self._cut
        self._cut = None
        tensors, grads = [hid], [leaf.grad]
        if self.training and loss.vae_rkl is not None:
            tensors.append(loss.vae_rkl)
            grads.append(None)
        if self.config.do_pred and loss.get("rating_mse") is not None:
            tensors.append(loss.rating_mse)
            grads.append(None)
        self._run_backward(lambda: torch.autograd.backward(tensors, grads))

    def backward_encoder(self, loss=None):
        """Third part (split_backward = 2): the utterance encoder and the encoder-side word lookup."""
        if self._cut_enc is None:
            return
        h, leaf = self._cut_enc
        self._cut_enc = None
        if leaf.grad is not None:
            self._run_backward(lambda: torch.autograd.backward([h], [leaf.grad]))

    def backward_rest(self, loss):
        self.backward_mid(loss)
        self.backward_encoder(loss)

    def forward(self, data_feed, mode=TEACH_FORCE, gen_type="greedy", eps=None):
        # forward-time sorts of the lookups' ids (ops.PRESORT) fork a background stream that the lookup's backward joins:
        # not when forward and backward are captured into different HIP graphs (split_backward)
        ops.PRESORT_OK = not self.split_backward
        ops.third_stream_reset()
        if self.training and self.config.dropout > 0:
            ops.rng_advance(self.x_embedding.weight.device)      # new dropout masks every step, also under graph replay
        prep_late = PREPARE_LATE and self.latent_stack == "dls"
        early_sn = self.latent_stack == "dls" and data_feed["user_ids"].is_cuda
        # (not with use_mutual: DeN_Auto's spectral pre-pass is several times the pyramid stack's, and beside the
        # recurrence it costs that configuration 5.00 -> 5.19 ms per step)
        head_side = (HEAD_SIDE and early_sn and prep_late and self.x_encoder.rnn.cell == "gru" and not self.split_backward
                     and mode == TEACH_FORCE and not self.use_mutual)
        if not head_side:
            rating_ids = data_feed["ratings"].ge(3).long()                                # :760
            self.__dict__["_rating_ids"] = rating_ids
        if data_feed["user_ids"].is_cuda and mode == TEACH_FORCE and PREPARE_ON_SIDE and not prep_late:
            self._prepare_on_side(data_feed)
        if early_sn:
            # all 105 power iterations (one launch group) beside the launches in front of the encoder recurrence, whose
            # launch joins the side stream; first read by the pyramids behind it
            if getattr(self, "_sn_modules", None) is None:
                self._sn_modules, self._sn_cache = L.spectral_modules(self), {}
            with ops.fork_side():
                ops.spectral_prepass(self._sn_modules, self.training, cache=self._sn_cache)
        if head_side:
            # only the word lookup stands in front of the encoder recurrence: the user / item / rating lookups feed the
            # prior path behind it, so they run on the side stream beside the recurrence (128 of the 256 CUs), and so does
            # the spectral pre-pass; joined in front of prior_fc
            words = ops.embedding(self.x_embedding.weight, data_feed["sentence_embeddings"], self.pad_id)
            with ops.fork_side():
                rating_ids = data_feed["ratings"].ge(3).long()                            # :760
                self.__dict__["_rating_ids"] = rating_ids
                user = ops.embedding(self.user_embedding.weight, data_feed["user_ids"])          # :758
                item = ops.embedding(self.item_embedding.weight, data_feed["item_ids"])          # :759
                rating = ops.embedding(self.rating_embedding.weight, rating_ids)                  # :760
            ops.FWD_RNN_BESIDE_SIDE = True
            try:
                enc_outs, enc_last = self.x_encoder(words)                                    # :765
            finally:
                ops.FWD_RNN_BESIDE_SIDE = False
            ops.join_side_work()
        else:
            user = ops.embedding(self.user_embedding.weight, data_feed["user_ids"])          # :758
            item = ops.embedding(self.item_embedding.weight, data_feed["item_ids"])          # :759
            rating = ops.embedding(self.rating_embedding.weight, rating_ids)                  # :760
            words = ops.embedding(self.x_embedding.weight, data_feed["sentence_embeddings"], self.pad_id)
            enc_outs, enc_last = self.x_encoder(words)                                        # :765
        h = enc_last.transpose(0, 1).contiguous().squeeze(1)                              # :766
        if (self.split_backward == 2 and torch.is_grad_enabled() and h.requires_grad
                and mode == TEACH_FORCE and not self.config.use_attention):
            # second cut of the data-parallel overlap schedule (the encoder states reach the decoder through h alone
            # without attention): behind it only the encoder's and the word table's gradients are still missing
            leaf = h.detach().requires_grad_()
            self._cut_enc, h = (h, leaf), leaf
        prior_h = ops.tanh(ops.linear(torch.cat([user, item, rating], -1), self.prior_fc.weight, self.prior_fc.bias))
        B = h.size(0)
        if self.latent_stack == "dls":
            if not early_sn:
                if getattr(self, "_sn_modules", None) is None:
                    self._sn_modules, self._sn_cache = L.spectral_modules(self), {}
                ops.spectral_prepass(self._sn_modules, self.training, cache=self._sn_cache)    # all 105 power iterations
            ops.join_side_work()
            if prep_late and data_feed["user_ids"].is_cuda and mode == TEACH_FORCE and PREPARE_ON_SIDE:
                self._prepare_on_side(data_feed)             # beside the latent stack; joined in front of the decoder
            (z_pri, pri_state), (z_rec, rec_state) = L.pyramid_pair(self.pri_encoder, prior_h,         # :3529
                                                                    self.rec_encoder, h)               # :3530
            if self.use_mutual:                                                            # :3089 (DeN_VAE)
                noise, gumbels = eps if isinstance(eps, tuple) else (eps, None)
                z, rkl = self.sup_decoder(z_pri, rec_state, pri_state, getattr(self, "restarted", False), noise, gumbels)
            else:
                z, rkl = self.sup_decoder(z_pri, z_rec, rec_state, pri_state,
                                          getattr(self, "restarted", False), noise=eps)    # :3535
            if not self.training:
                rkl = None                                                                 # :3571
            dec_init = self.x_init_connector(z).unsqueeze(0)                                  # :3473
        else:
            mu_q, lv_q = self.rec_head(h)
            mu_p, lv_p = self.pri_head(prior_h)
            if self.training:
                if eps is None:
                    eps = torch.randn(B, mu_q.size(1), device=h.device)                   # :197
                z, kl_row = ops.reparam_kl(mu_q, lv_q, mu_p, lv_p, eps, mode=0)           # :191-203 + criterions.py:145-155
                rkl = ops.mean(kl_row)
            else:
                z, rkl = mu_p, None                                                        # eval: condition on the prior mean
            dec_init = self.x_init_connector(z)                                            # :711
        attn_ctx = enc_outs if self.config.use_attention else None
        feature = (ops.embedding(self.feature_embedding.weight, data_feed["feature_ids"])
                   if self.use_feature else None)                                          # :762-763
        if prep_late:
            ops.join_side_work()
        dec_outs, dec_last, dec_ctx = self.x_decoder(B, data_feed["input_ids"], dec_init, mode=mode,
                                                     gen_type=gen_type, beam_size=self.config.beam_size,
                                                     attn_context=attn_ctx, latent_variable=feature)   # :712
        if self.split_backward and torch.is_grad_enabled() and hasattr(dec_outs, "token_nll"):
            # data-parallel overlap (dp.FlatGradReducer.start / finish): cut the autograd graph at the decoder states, so
            # that backward_vocab() yields the vocabulary projection's gradient -- a third of the dense bytes -- before
            # the recurrences and the latent stack run backward (backward_rest)
            leaf = dec_outs.hid.detach().requires_grad_()
            self._cut, dec_outs.hid = (dec_outs.hid, leaf), leaf
        if not self.backLoss:                                                              # generation: :813-817
            if self.config.do_pred:
                return dec_ctx, data_feed["ans_ids"], self._rating(user, item)
            return dec_ctx, data_feed["ans_ids"]
        vae_nll = self.nll_loss(dec_outs, data_feed["ans_ids"])                            # :790
        vae_ppl = self.ppl(dec_outs, data_feed["ans_ids"])                                 # :791
        pack = Pack(vae_zkl=None, vae_nll=vae_nll, vae_PPL=vae_ppl, vae_rnll=None, vae_rkl=rkl)
        if self.config.do_pred:
            diff = self._rating(user, item) - data_feed["ratings"]                         # main_pretrain.py:891-893
            pack["rating_mse"] = ops.mean(diff * diff)
        return pack

    def _prepare_on_side(self, data_feed):
        """What the decoder half of the step needs and the inputs of the step already determine, made on the side stream
        beside the latent stack's forward (joined in front of the decoder; PEVAE_PREPARE_LATE=0 or the flat Gaussian stack:
        beside the embedding / encoder launches, joined by the encoder recurrence's launch): the bf16 operand copies of
        the vocabulary projection and of the decoder's recurrence weights, and -- for the teacher-forced fused path -- the
        targets in operand row order with their live-row partition.  Consumed once by ops.pack_matrix /
        DeferredLogProbs.token_nll; anything left over is dropped at the next call."""
        ops.clear_prepared()
        dec = self.x_decoder
        rnn = dec.rnn
        if rnn.cell != "gru" or rnn.num_layers != 1 or rnn.bidirectional:
            return
        labels = data_feed["ans_ids"]
        B, Ld = labels.shape[0], labels.shape[1]
        H = rnn.hidden_size
        Hk, _K3, Bpad, _NG = ops.gru_geometry(B, H)
        w_ih, w_hh, _b1, _b2 = rnn.layer_weights(0, 0)
        with ops.fork_side():
            ops.prepack_weights([(dec._out_weights()[0], ops.round_up(H, 64)), (w_hh, Hk),
                                 (w_ih, ops.round_up(w_ih.shape[1], 64))])
            if not (self.config.use_attention) and torch.is_grad_enabled():
                ops.precompact_targets(labels, (Ld, B, Bpad), self.pad_id)
            if not self.use_feature and self.training:
                dec.prepare_teacher_forced(data_feed["input_ids"])        # input embedding + dropout + input projection
            if torch.is_grad_enabled() and self.training:
                # W_hh^T / W_ih^T of both recurrences for the backward pass (GRU, one layer, one direction: as above)
                enc = self.x_encoder.rnn if hasattr(self.x_encoder, "rnn") else None
                for r in (rnn, enc):
                    if r is None or r.cell != "gru" or r.num_layers != 1 or r.bidirectional:
                        continue
                    wi, wh, _b1, _b2 = r.layer_weights(0, 0)
                    Hr = r.hidden_size
                    _hk, k3, _bp, _ng = ops.gru_geometry(B, Hr)
                    ops.pretranspose_weights([(wh, 3 * Hr, Hr, Hr, k3), (wi, 3 * Hr, wi.shape[1], wi.shape[1], k3)])
            if torch.is_grad_enabled():
                # the id sorts of the table gradients (the lookups' backward passes are left with their segmented sums)
                ops.presort_lookup(data_feed["sentence_embeddings"], self.x_embedding.weight, "words")
                ops.presort_lookup(data_feed["user_ids"], self.user_embedding.weight, "user")
                ops.presort_lookup(data_feed["item_ids"], self.item_embedding.weight, "item")
                ops.presort_lookup(self.__dict__["_rating_ids"], self.rating_embedding.weight, "rating")
                if not self.use_feature and self.training:
                    ops.presort_lookup(data_feed["input_ids"], dec.embedding.weight, "dec_in")

    def _rating(self, user, item):
        x = torch.cat([user, item], -1)
        x = ops.linear(x, self.dnn[0].weight, self.dnn[0].bias)
        return ops.linear(ops.tanh(x), self.dnn[1].weight, self.dnn[1].bias).squeeze(-1)


def synthetic_batch(config, batch_size, seed, device="cpu"):
    """Synthetic batch in the format of YELPDataLoader.batcher.__call__ (data_loaders.py:704-715):
    BOS, len-2 Zipf-distributed word ids in [6, V), EOS, PAD to L; ans_ids = input_ids shifted left.  BOS / EOS are
    config.bos_id / config.eos_id (4 / 5 by default: where a real corpus puts them, see the top of this file)."""
    g = torch.Generator().manual_seed(seed)
    B, L, V = batch_size, config.max_seq_len, config.vocab_size
    lens = torch.randint(min(8, L), L + 1, (B,), generator=g)
    ranks = torch.arange(1, V - N_SPECIAL + 1, dtype=torch.float64)
    probs = (1.0 / ranks ** 1.1)
    words = torch.multinomial(probs / probs.sum(), B * L, replacement=True, generator=g).view(B, L) + N_SPECIAL
    pos = torch.arange(L).unsqueeze(0)
    ids = torch.where(pos < lens.unsqueeze(1), words, torch.zeros_like(words))
    ids[:, 0] = getattr(config, "bos_id", BOS_ID)
    ids[torch.arange(B), lens - 1] = getattr(config, "eos_id", EOS_ID)
    ans = torch.cat([ids[:, 1:], torch.zeros(B, 1, dtype=torch.long)], 1)
    user = torch.randint(0, config.n_user, (B,), generator=g)
    item = torch.randint(0, config.n_item, (B,), generator=g)
    feed = {
        "user_ids": user, "item_ids": item,
        "feature_ids": torch.randint(0, max(1, getattr(config, "n_feature", 0)), (B,), generator=g),
        "sentence_embeddings": ids, "attention_masks": (ids != 0).long(),
        "ratings": torch.randint(0, 5, (B,), generator=g).float(), "input_ids": ids.clone(), "ans_ids": ans,
        "rd_ids": torch.randn(B, generator=g), "ui_ids": user + 1000000000 * item,
    }
    return {k: v.to(device).contiguous() for k, v in feed.items()}
