"""CPU oracle of the composed DLS_VAE training step (TEST INFRASTRUCTURE; see oracle/__init__.py).

Same composition and state_dict keys as pevae_amd/dls_vae.py, built from the oracle restatements of the
reference modules (oracle/ref_modules.py): data flow of FS_VAE.forward (dialog_models.py:755-818), loss /
Pack contract of DLS_VAE (dialog_models.py:3451-3465, :3569).  Also the `cpu_baseline` leg of bench.py
(stock ATen fp32 kernels on the host cores, the same ones the reference modules call).
"""
import torch
import torch.nn as nn

from . import ref_modules as R

PAD_ID, UNK_ID, BOS_ID, EOS_ID = 0, 1, 4, 5      # [<pad>, <unk>, <sys>, <usr>] + words (corpora.py:649); dialog_models.py:455-456


class RefDLSVAE(nn.Module):
    def __init__(self, config):
        super().__init__()
        c = self.config = config
        H, E, Z = c.dec_cell_size, c.embed_size, c.latent_size
        self.emb_size = H // 2
        self.pad_id, self.unk_id = getattr(c, "pad_id", PAD_ID), getattr(c, "unk_id", UNK_ID)
        self.go_id, self.eos_id = getattr(c, "bos_id", BOS_ID), getattr(c, "eos_id", EOS_ID)
        self.x_embedding = nn.Embedding(c.vocab_size, E, padding_idx=self.pad_id)
        self.user_embedding = nn.Embedding(c.n_user, self.emb_size)
        self.item_embedding = nn.Embedding(c.n_item, self.emb_size)
        self.rating_embedding = nn.Embedding(5, self.emb_size)
        self.x_encoder = R.EncoderRNN(E, H, dropout_p=c.dropout, rnn_cell=c.rnn_cell, variable_lengths=False)
        self.use_feature = bool(getattr(c, "use_feature", False))
        if self.use_feature:                                                             # :488, :762-763
            self.feature_embedding = nn.Embedding(c.n_feature, self.emb_size)
        self.x_decoder = R.DecoderRNN(c.vocab_size, c.max_seq_len, E + (self.emb_size if self.use_feature else 0), H,
                                      self.go_id, self.eos_id, self.unk_id,
                                      n_layers=c.num_layer, rnn_cell=c.rnn_cell, input_dropout_p=c.dropout,
                                      dropout_p=c.dropout, use_attention=c.use_attention, attn_mode=c.attn_type,
                                      attn_size=H, use_gpu=False, embedding=self.x_embedding)
        self.latent_stack = getattr(c, "latent_stack", "gauss")
        self.use_mutual = bool(getattr(c, "use_mutual", False)) and self.latent_stack == "dls"
        if self.latent_stack == "dls":                                                   # dialog_models.py:3401-3417
            self.rec_encoder = R.DlsPyramid(H, H // 16, c)
            self.pri_encoder = R.DlsPyramid(H, H // 16, c, wrap=True)
            if self.use_mutual:                                                          # DeN_VAE, :2917-2931
                self.sup_decoder = R.DeNAuto(H // 32, H // 16, c)
                self.x_init_connector = nn.Sequential(R.Block(H // 8, H // 4, H // 4), R.Block(H // 4, H // 2, H // 2),
                                                      R.Block(H // 2, H, H))
            else:
                self.sup_decoder = R.DLSAuto(H // 32, H // 8, c)
                self.x_init_connector = nn.Sequential(R.Block(H // 4, H // 2, H // 2), R.Block(H // 2, H, H))
        self.prior_fc = nn.Linear(3 * self.emb_size, H)
        if self.latent_stack == "gauss":
            self.rec_head = R.Hidden2Gaussian(H, Z)
            self.pri_head = R.Hidden2Gaussian(H, Z)
            self.x_init_connector = R.LinearConnector(Z, H, c.rnn_cell == "lstm")
        if c.do_pred:
            self.dnn = nn.Sequential(nn.Linear(2 * self.emb_size, H), nn.Linear(H, 1))
        self.nll_loss = R.NLLEntropy(self.pad_id, c)
        self.ppl = R.Perplexity(self.pad_id, c)
        self.kl_loss = R.NormKLLoss(unit_average=False)

    def use_aten_rnn(self, on=True):
        """Timing leg: run both GRUs through stock torch.nn.GRU (the kernels the reference itself calls)."""
        self.x_encoder.rnn.use_aten = on
        self.x_decoder.rnn.use_aten = on
        return self

    def valid_loss(self, loss):
        total = loss.vae_nll
        if self.training and loss.vae_rkl is not None:
            total = total + loss.vae_rkl
        if self.config.do_pred and loss.get("rating_mse") is not None:
            total = total + loss.rating_mse
        return total

    def forward(self, feed, mode=R.TEACH_FORCE, gen_type="greedy", eps=None):
        user = self.user_embedding(feed["user_ids"])
        item = self.item_embedding(feed["item_ids"])
        rating = self.rating_embedding(feed["ratings"].ge(3).long())
        enc_outs, enc_last = self.x_encoder(self.x_embedding(feed["sentence_embeddings"]))
        h = enc_last.transpose(0, 1).contiguous().squeeze(1)
        prior_h = torch.tanh(self.prior_fc(torch.cat([user, item, rating], -1)))
        B = h.size(0)
        if self.latent_stack == "dls":
            z_pri, pri_state = self.pri_encoder(prior_h)
            z_rec, rec_state = self.rec_encoder(h)
            if self.use_mutual:                                                            # :3089
                noise, gumbels = eps if isinstance(eps, tuple) else (eps, None)
                z, rkl = self.sup_decoder(z_pri, rec_state, pri_state, getattr(self, "restarted", False), noise, gumbels)
            else:
                z, rkl = self.sup_decoder(z_pri, z_rec, rec_state, pri_state, getattr(self, "restarted", False), noise=eps)
            if not self.training:
                rkl = None
            dec_init = self.x_init_connector(z).unsqueeze(0)
        else:
            mu_q, lv_q = self.rec_head(h)
            mu_p, lv_p = self.pri_head(prior_h)
            if self.training:
                if eps is None:
                    eps = torch.randn(B, mu_q.size(1))
                z = R.reparameterization(mu_q, lv_q, eps)
                rkl = self.kl_loss(mu_q, lv_q, mu_p, lv_p)
            else:
                z, rkl = mu_p, None
            dec_init = self.x_init_connector(z)
        ctx = enc_outs if self.config.use_attention else None
        feature = self.feature_embedding(feed["feature_ids"]) if self.use_feature else None
        dec_outs, _, _ = self.x_decoder(B, feed["input_ids"], dec_init, mode=mode, gen_type=gen_type,
                                        beam_size=self.config.beam_size, attn_context=ctx, latent_variable=feature)
        pack = R.Pack(vae_zkl=None, vae_nll=self.nll_loss(dec_outs, feed["ans_ids"]),
                      vae_PPL=self.ppl(dec_outs, feed["ans_ids"]), vae_rnll=None, vae_rkl=rkl)
        if self.config.do_pred:
            x = torch.cat([user, item], -1)
            pred = self.dnn[1](torch.tanh(self.dnn[0](x))).squeeze(-1)
            pack["rating_mse"] = ((pred - feed["ratings"]) ** 2).mean()
        return pack
