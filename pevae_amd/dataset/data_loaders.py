"""Yelp data path of the DLS_VAE step: vocabulary build, batch collation and the epoch iterator.

Mirrors the reference's interface for this path --
  * `build_vocab`          <- YelpCorpus._build_vocab         (models/dataset/corpora.py:624-651)
  * `YELPDataLoader`       <- YELPDataLoader                  (models/dataset/data_loaders.py:613-634)
  * `YELPDataLoader.batcher` <- YELPDataLoader.batcher        (models/dataset/data_loaders.py:636-715)
  * `epoch_init/next_batch`<- DataLoader.epoch_init/next_batch (models/dataset/dataloader_bases.py:30-55) over the
    torch DataLoader the model builds (dialog_models.py:528: batch_size, sampler, collate_fn=batcher, 1 worker)
-- with one design change: the reference's batcher builds Python lists and calls `.to(device)` ten times inside the
worker; here a batch is collated into a few numpy arrays, staged in reused page-locked buffers and sent with
asynchronous H2D copies on the consumer's stream, so the copy of batch k+1 overlaps step k.  A row is the corpus tuple
(user, item, feature, tokens, rating in 1..5, rd) of corpora.py:609-622.  The corpus pickles themselves are not part of
the path (none ship with the reference)."""
import re
from collections import Counter

import numpy as np
import torch

PAD, UNK, BOS, EOS = "<pad>", "<unk>", "<s>", "</s>"        # corpora.py:533-536
SYS, USR = "<sys>", "<usr>"                                  # corpora.py:24-25


_CHAT_TOKEN = re.compile(r"\w+|<sil>|[^\w\s]+")


def chat_tokenize(text):
    """utils.get_chat_tokenize (utils.py:241-242): nltk.RegexpTokenizer(r'\\w+|<sil>|[^\\w\\s]+').tokenize, which with
    gaps=False is re.findall of that pattern (nltk itself is not needed)."""
    return _CHAT_TOKEN.findall(text)


def process_reviews(data, tokenize=chat_tokenize, rd=None):
    """yelp._process_reivews (corpora.py:609-622): raw review tuples (user, item, feature, text, ..., rating) -> corpus
    rows (user, item, feature, [BOS] + tokens + [EOS], rating, rd).  rd is the per-sample N(0,1) draw the reference
    attaches with torch.randn(1).item() (:621); pass a sequence to fix it, default draws it the same way."""
    res = []
    for i, d in enumerate(data):
        sentence = [BOS] + tokenize(d[3]) + [EOS]
        res.append((d[0], d[1], d[2], sentence, d[-1], torch.randn(1).item() if rd is None else rd[i]))
    return res


def build_vocab(corpora, max_vocab_cnt):
    """corpora: iterables of rows (train, valid, test, pretrain in the reference's order); words counted over all of
    them, most frequent first (ties: first seen first, as collections.Counter.most_common does), cut at max_vocab_cnt.
    -> (vocab list, rev_vocab dict, stats) with vocab = [PAD, UNK, SYS, USR] + words (corpora.py:644-646)."""
    all_words = []
    for corpus in corpora:
        for example in corpus:
            all_words.extend(example[3])
    vocab_count = Counter(all_words).most_common()
    raw_vocab_size = len(vocab_count)
    discard_wc = int(np.sum([c for _, c in vocab_count[max_vocab_cnt:]]))
    vocab_count = vocab_count[0:max_vocab_cnt]
    vocab = [PAD, UNK, SYS, USR] + [t for t, _ in vocab_count]
    rev_vocab = {t: idx for idx, t in enumerate(vocab)}
    stats = dict(raw_vocab_size=raw_vocab_size, vocab_size=len(vocab_count),
                 cut_off=vocab_count[-1][1] if vocab_count else 0,
                 oov_rate=float(discard_wc) / max(len(all_words), 1))
    return vocab, rev_vocab, stats


class YELPDataLoader(object):
    KEYS_LONG = ("user_ids", "item_ids", "feature_ids", "ui_ids")
    KEYS_SEQ = ("sentence_embeddings", "attention_masks", "input_ids", "ans_ids")

    def __init__(self, name, data, config, max_utt_len=-1):
        self.name, self.data, self.config = name, data, config
        self.fix_batch = False
        self.indexes = list(range(len(data)))
        self.max_utt_size = config.max_seq_len
        self.data_size = len(data)
        self.rev_vocab = None
        self.sampler = None
        self.collate = None
        self.ptr, self.num_batch, self.order = 0, 0, None
        self.rng = np.random.RandomState(getattr(config, "seed", 0))

    def attach(self, rev_vocab, device):
        """What the model does at dialog_models.py:524-528: hand the loader its vocabulary and build the collate."""
        self.rev_vocab = rev_vocab
        self.collate = self.batcher(rev_vocab, self.config.max_seq_len, device)
        return self

    class batcher(object):
        def __init__(self, rev_vocab, max_seq_len, device):
            self.rev_vocab = rev_vocab
            self.max_seq_len = max_seq_len
            self.device = torch.device(device)
            self.unk = rev_vocab[UNK]
            self._ring, self._slot = [], 0                   # page-locked staging sets, reused round-robin

        @staticmethod
        def pad_to(max_len, tokens, do_pad=True):            # data_loaders.py:641-648
            if len(tokens) >= max_len:
                return tokens[0:max_len - 1] + [tokens[-1]]
            elif do_pad:
                return tokens + [0] * (max_len - len(tokens))
            return tokens

        def collate_host(self, rows):
            """The batch as numpy arrays (data_loaders.py:649-715): ids looked up with <unk> for OOV, input padded /
            truncated to max_seq_len keeping the last token, answers = input shifted left, mask = ones for all but
            the last token, ratings - 1, ui = user + 1e9 * item."""
            B, L = len(rows), self.max_seq_len
            out = {k: np.zeros(B, dtype=np.int64) for k in YELPDataLoader.KEYS_LONG}
            out.update({k: np.zeros((B, L), dtype=np.int64) for k in YELPDataLoader.KEYS_SEQ})
            out["ratings"] = np.zeros(B, dtype=np.float32)
            out["rd_ids"] = np.zeros(B, dtype=np.float32)
            get, unk = self.rev_vocab.get, self.unk
            for i, row in enumerate(rows):
                if row[4] <= 0 or row[4] > 5:
                    raise ValueError("YELPDataLoader.batcher: illegal rating %r in row %d" % (row[4], i))   # :695-697 exits
                out["user_ids"][i], out["item_ids"][i], out["feature_ids"][i] = row[0], row[1], row[2]
                out["ui_ids"][i] = row[0] + 1000000000 * row[1]
                out["ratings"][i] = row[4] - 1
                out["rd_ids"][i] = row[5]
                ids = [get(w, unk) for w in row[3]]
                out["ans_ids"][i] = self.pad_to(L, ids[1:])
                out["input_ids"][i] = self.pad_to(L, ids)
                out["attention_masks"][i] = self.pad_to(L, [1] * (len(ids) - 1))
            out["sentence_embeddings"] = out["input_ids"]
            return out

        def __call__(self, rows):
            host = self.collate_host(rows)
            if self.device.type != "cuda":
                return {k: torch.from_numpy(v.copy()).contiguous() for k, v in host.items()}
            if len(self._ring) < 3:
                self._ring.append(({}, torch.cuda.Event()))
                slot = self._ring[-1]
            else:
                slot = self._ring[self._slot % 3]
                slot[1].synchronize()                        # the copies that last used this staging set are done
            self._slot += 1
            stage, ev = slot
            batch = {}
            for k, v in host.items():
                buf = stage.get(k)
                if buf is None or buf.shape != v.shape:
                    buf = stage[k] = torch.empty(v.shape, dtype=torch.from_numpy(v).dtype, pin_memory=True)
                buf.numpy()[...] = v
                batch[k] = buf.to(self.device, non_blocking=True)
            ev.record()
            return batch

    def epoch_init(self, config, shuffle=True, verbose=True):
        """dataloader_bases.py:30-49: num_batch = data_size // batch_size; a shuffled pass over the data (the
        reference's sampler), the last short batch included like torch's DataLoader without drop_last."""
        if self.collate is None:
            raise RuntimeError("YELPDataLoader.epoch_init: call attach(rev_vocab, device) first")
        self.batch_size = config.batch_size
        self.num_batch = self.data_size // config.batch_size
        self.order = list(self.indexes)
        if shuffle:
            self.rng.shuffle(self.order)
        self.ptr = 0
        return self.num_batch

    def next_batch(self):
        if self.order is None or self.ptr >= self.data_size:
            return None
        sel = self.order[self.ptr:self.ptr + self.batch_size]
        self.ptr += self.batch_size
        return self.collate([self.data[i] for i in sel])
