"""Yelp data path (pevae_amd/dataset/data_loaders.py) against the list-by-list restatement of the reference's collation
(oracle/ref_batcher.py): vocabulary build, batcher output, epoch iteration; and the engine's LossManager report."""
import random
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ref_batcher as R
from pevae_amd.dataset import YELPDataLoader, build_vocab, BOS, EOS


def make_rows(n, seed, n_words=120, max_len=30):
    rnd = random.Random(seed)
    words = ["w%d" % i for i in range(n_words)]
    rows = []
    for _ in range(n):
        k = rnd.choice([0, 1, 2, 5, 9, 10, 11, 12, 20, max_len])
        toks = [BOS] + [rnd.choice(words[: rnd.choice([5, 40, n_words])]) for _ in range(k)] + [EOS]
        rows.append((rnd.randrange(50), rnd.randrange(40), rnd.randrange(7), toks, rnd.randint(1, 5), rnd.gauss(0, 1)))
    return rows


def test_build_vocab_matches_reference_order():
    tr, va, te, pre = make_rows(200, 1), make_rows(30, 2), make_rows(30, 3), make_rows(10, 4)
    vocab, rev, stats = build_vocab((tr, va, te, pre), 60)
    rv, rr = R.build_vocab(tr, va, te, pre, 60)
    assert vocab == rv and rev == rr
    assert vocab[:4] == ["<pad>", "<unk>", "<sys>", "<usr>"] and len(vocab) == 64
    assert 0.0 < stats["oov_rate"] < 1.0 and stats["raw_vocab_size"] == 122


@pytest.mark.parametrize("L", [8, 12, 50])
def test_batcher_matches_reference_collation(L):
    rows = make_rows(64, 7)
    _, rev, _ = build_vocab((rows[:40],), 50)              # the other rows bring out-of-vocabulary words
    got = YELPDataLoader.batcher(rev, L, "cpu")(rows)
    want = R.batch(rows, rev, L)
    assert set(got) == set(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k
    assert got["input_ids"].is_contiguous()
    if L < 30:
        assert [r for r in rows if len(r[3]) >= L], "the fixture must hold sentences that need truncation"


def test_batcher_rejects_illegal_rating():
    rows = make_rows(4, 9)
    rows[2] = rows[2][:4] + (0,) + rows[2][5:]
    _, rev, _ = build_vocab((rows,), 50)
    with pytest.raises(ValueError):
        YELPDataLoader.batcher(rev, 12, "cpu")(rows)


def test_epoch_visits_every_row_once():
    rows = make_rows(70, 11)
    _, rev, _ = build_vocab((rows,), 200)
    cfg = Namespace(max_seq_len=12, batch_size=16, fix_batch=False, seed=3)
    dl = YELPDataLoader("train", rows, cfg).attach(rev, "cpu")
    with pytest.raises(RuntimeError):
        YELPDataLoader("x", rows, cfg).epoch_init(cfg)
    seen = []
    for epoch in range(2):
        assert dl.epoch_init(cfg, shuffle=True) == 70 // 16
        sizes = []
        while True:
            b = dl.next_batch()
            if b is None:
                break
            sizes.append(b["user_ids"].shape[0])
            seen.append(b["rd_ids"])
        assert sizes == [16, 16, 16, 16, 6]
    allrd = torch.cat(seen).numpy()
    want = np.sort(np.array([r[5] for r in rows] * 2, dtype=np.float32))
    assert np.array_equal(np.sort(allrd), want)
    assert not np.array_equal(allrd[:70], allrd[70:])      # reshuffled between epochs


def test_loss_manager_report_format():
    from pevae_amd.engine import LossManager
    lm = LossManager()
    for v in (2.0, 4.0):
        lm.add_loss({"vae_nll": torch.tensor(v), "vae_rkl": torch.tensor(v / 4), "flag": True, "n": 3, "none": None})
    assert lm.avg_loss() == 3.0
    d = lm.return_dict()
    assert d["vae_nll"] == 3.0 and abs(d["vae_PPL"] - float(np.exp(3.0))) < 1e-4 and d["vae_rkl"] == 0.75
    assert lm.pprint("Train", prefix="1/2") == "1/2: Train vae_nll 3.000 PPL 20.086 vae_rkl 0.750"
    assert lm.pprint("Train", window=1).startswith("Train vae_nll 4.000")
    lm.clear()
    assert lm.return_dict() == {}


def _golden_data_path():
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "yelp_data_path.json")) as fh:
        return json.load(fh)


def test_data_path_matches_reference_code_golden():
    """tests/golden/yelp_data_path.json was produced by the reference's OWN yelp._process_reivews / _build_vocab
    (corpora.py:609-651) and YELPDataLoader.batcher (data_loaders.py:636-715) on synthetic reviews
    (oracle/make_golden.py gen_yelp_data_path).  Both the product path and the oracle restatement must reproduce the
    processed rows, the vocabulary and every collated batch exactly."""
    from pevae_amd.dataset import process_reviews
    g = _golden_data_path()
    rows = {}
    for name, raw in g["raw"].items():
        want = g["rows"][name]
        got = process_reviews([tuple(r) for r in raw], rd=[w[5] for w in want])
        assert [list(r) for r in got] == want, name
        rows[name] = got
    order = ("train", "valid", "test", "pretrain")
    vocab, rev, _ = build_vocab(tuple(rows[k] for k in order), g["max_vocab_cnt"])
    assert vocab == g["vocab"] and rev["<unk>"] == g["unk_id"]
    rv, rr = R.build_vocab(*(rows[k] for k in order), g["max_vocab_cnt"])
    assert rv == g["vocab"] and rr == rev
    assert any(w not in rev for r in rows["train"] for w in r[3]), "the fixture must contain out-of-vocabulary words"
    for b in g["batches"]:
        sel = [rows["train"][i] for i in b["rows"]]
        got = YELPDataLoader.batcher(rev, b["max_seq_len"], "cpu")(sel)
        ora = R.batch(sel, rev, b["max_seq_len"])
        assert set(got) == set(b["out"]) == set(ora)
        for k, want in b["out"].items():
            want_t = torch.tensor(want, dtype=got[k].dtype)
            assert torch.equal(got[k], want_t), (k, b["max_seq_len"])
            assert torch.equal(ora[k], want_t) and ora[k].dtype == got[k].dtype, (k, b["max_seq_len"])
    assert any(len(rows["train"][i][3]) > b["max_seq_len"] for b in g["batches"] for i in b["rows"])   # truncation exercised


@pytest.mark.gpu
def test_data_path_cuda_batcher_matches_reference_code_golden():
    """The CUDA branch of the batcher (pinned staging ring of three sets, event-guarded reuse, non-blocking H2D copies;
    data_loaders.py:649-715 moved the batch with .to(device) inside the worker) against the same golden: every batch of
    the fixture several times over through ONE batcher object -- each staging set is reused at least twice, with batches of
    different shapes in between -- and every tensor that arrives on the device is bit-identical to what the reference's own
    batcher produced.  The device tensors are checked only after ALL calls have been issued, so a staging set that was
    overwritten before its copy had finished would show."""
    assert torch.cuda.is_available()
    from pevae_amd.dataset import process_reviews
    g = _golden_data_path()
    rows = {name: process_reviews([tuple(r) for r in raw], rd=[w[5] for w in g["rows"][name]]) for name, raw in g["raw"].items()}
    order = ("train", "valid", "test", "pretrain")
    _, rev, _ = build_vocab(tuple(rows[k] for k in order), g["max_vocab_cnt"])
    by_len = {}
    for b in g["batches"]:
        by_len.setdefault(b["max_seq_len"], []).append(b)
    calls = 0
    for L, batches in by_len.items():
        collate = YELPDataLoader.batcher(rev, L, "cuda")
        issued = []
        for rep in range(3):                               # >= 7 calls per batcher whenever the fixture has 3+ batches of a length
            for b in batches:
                issued.append((b, collate([rows["train"][i] for i in b["rows"]])))
        while len(issued) < 7:
            b = batches[len(issued) % len(batches)]
            issued.append((b, collate([rows["train"][i] for i in b["rows"]])))
        assert len(collate._ring) == 3 and collate._slot == len(issued) >= 7
        torch.cuda.synchronize()
        for b, got in issued:
            assert set(got) == set(b["out"])
            for k, want in b["out"].items():
                assert got[k].is_cuda and got[k].is_contiguous()
                want_t = torch.tensor(want, dtype=got[k].dtype)
                assert torch.equal(got[k].cpu(), want_t), (k, L)
        calls += len(issued)
    assert calls >= 7
