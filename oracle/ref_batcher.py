"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md: nothing in the product path imports this).
Plain-Python restatement of the reference's Yelp collation and vocabulary build, list by list as the reference does it:
  batch()        <- YELPDataLoader.batcher.__call__   models/dataset/data_loaders.py:649-715
  pad_to()       <- YELPDataLoader.batcher.pad_to     models/dataset/data_loaders.py:641-648
  build_vocab()  <- YelpCorpus._build_vocab           models/dataset/corpora.py:624-651
Pinned by tests/golden/yelp_data_path.json, which oracle/make_golden.py gen_yelp_data_path produced by running the
reference's own corpus / loader code (imported with the nltk and numpy.lib.type_check stubs) on a synthetic review file:
tests/test_data_path.py compares this restatement and pevae_amd.dataset with it list by list."""
from collections import Counter

import torch


def pad_to(max_len, tokens, do_pad=True):
    if len(tokens) >= max_len:
        return tokens[0:max_len - 1] + [tokens[-1]]
    elif do_pad:
        return tokens + [0] * (max_len - len(tokens))
    else:
        return tokens


def batch(rows, rev_vocab, max_seq_len):
    user_ids, item_ids, feature_ids, sentence_embeddings, input_ids = [], [], [], [], []
    ratings, ans_ids, rd_ids, at_masks, ui_ids = [], [], [], [], []
    pad_length = max_seq_len
    for idx, row in enumerate(rows):
        user_ids.append(rows[idx][0])
        item_ids.append(rows[idx][1])
        ui_ids.append(rows[idx][0] + 1000000000 * rows[idx][1])
        feature_ids.append(rows[idx][2])
        ratings.append(rows[idx][4] - 1)
        input_id_now, attention_mask = [], []
        for word in rows[idx][3]:
            try:
                input_id_now.append(rev_vocab[word])
            except KeyError:
                input_id_now.append(rev_vocab['<unk>'])
            attention_mask.append(1)
        ans_id = pad_to(pad_length, input_id_now[1:])
        input_id_now = pad_to(pad_length, input_id_now)
        attention_mask = pad_to(pad_length, attention_mask[:-1])
        input_ids.append(input_id_now)
        ans_ids.append(ans_id)
        rd_ids.append(rows[idx][5])
        at_masks.append(attention_mask)
        sentence_embeddings.append(input_id_now)
    return {
        'user_ids': torch.tensor(user_ids, dtype=torch.long), 'item_ids': torch.tensor(item_ids, dtype=torch.long),
        'feature_ids': torch.tensor(feature_ids, dtype=torch.long),
        'sentence_embeddings': torch.tensor(sentence_embeddings, dtype=torch.long),
        'attention_masks': torch.tensor(at_masks, dtype=torch.long), 'ratings': torch.tensor(ratings, dtype=torch.float32),
        'input_ids': torch.tensor(input_ids, dtype=torch.long), 'ans_ids': torch.tensor(ans_ids, dtype=torch.long),
        'rd_ids': torch.tensor(rd_ids, dtype=torch.float), 'ui_ids': torch.tensor(ui_ids, dtype=torch.long),
    }


def build_vocab(train, valid, test, pretrain, max_vocab_cnt):
    all_words = []
    for corpus in (train, valid, test, pretrain):
        for example in corpus:
            all_words.extend(example[3])
    vocab_count = Counter(all_words).most_common()
    vocab_count = vocab_count[0:max_vocab_cnt]
    vocab = ['<pad>', '<unk>', '<sys>', '<usr>'] + [t for t, cnt in vocab_count]
    rev_vocab = {t: idx for idx, t in enumerate(vocab)}
    return vocab, rev_vocab
