"""Host data path: batching semantics of util.py:17-65 / data.py:11-117: hand-derived known answers,
the reference's own function bodies run on seeded inputs (tests/golden/make_data_golden.py -> reference_data.json) and
hypothesis properties against an independent statement of the rule."""
import numpy as np

from zero_amd.data import batch_indexer, token_indexer, Dataset
from zero_amd.vocab import Vocab


def test_batch_indexer():
    assert batch_indexer(7, 3) == [[0, 1, 2], [3, 4, 5], [6]]
    assert batch_indexer(6, 3) == [[0, 1, 2], [3, 4, 5]]
    assert batch_indexer(2, 5) == [[0, 1]]


def test_token_indexer_closes_before_the_budget():
    # util.py:48-52: 64-token sentences under token_size=4096 -> 63 per batch
    b = token_indexer([(64, 64)] * 200, 4096)
    assert [len(x) for x in b] == [63, 63, 63, 11]
    assert b[0][0] == 0 and b[1][0] == 63 and b[-1][-1] == 199
    # a single over-long sample is its own batch
    assert token_indexer([(10, 10), (5000, 3), (10, 10)], 100) == [[0], [1], [2]]
    # the longer side decides
    b = token_indexer([(2, 10)] * 25, 100)
    assert [len(x) for x in b] == [9, 9, 7]
    assert token_indexer([], 10) == []


def _write(tmp_path, n=57):
    rng = np.random.default_rng(0)
    words = ["w%d" % i for i in range(30)]
    src, tgt = tmp_path / "s.txt", tmp_path / "t.txt"
    with open(src, "w") as fs, open(tgt, "w") as ft:
        for _ in range(n):
            fs.write(" ".join(rng.choice(words, rng.integers(1, 12))) + "\n")
            ft.write(" ".join(rng.choice(words, rng.integers(1, 12))) + "\n")
    voc = tmp_path / "v.txt"
    voc.write_text("\n".join(words[:20]) + "\n")
    return str(src), str(tgt), str(voc)


def test_dataset_batches_cover_everything_once(tmp_path):
    src, tgt, voc = _write(tmp_path)
    v = Vocab(voc)
    for mode, size in (("batch", 8), ("token", 60)):
        ds = Dataset(src, tgt, v, v, max_len=8, batch_or_token=mode, data_leak_ratio=0.5)
        seen = []
        np.random.seed(1)
        for data in ds.batcher(size, buffer_size=20, shuffle=True, train=False):
            s, t = data['src'], data['tgt']
            assert s.dtype == np.int32 and s.shape[0] == t.shape[0] == len(data['index'])
            assert s.shape[1] <= 8 and t.shape[1] <= 8           # max_len caps the matrix (eos may be cut)
            assert (s[:, 0] != 0).all()
            if mode == "token":
                assert max((s > 0).sum(), (t > 0).sum()) < 60 or s.shape[0] == 1
            seen += data['index']
        assert sorted(seen) == list(range(57))
        assert ds.leak_buffer == []


def test_training_mode_leaks_small_tail_batches(tmp_path):
    src, tgt, voc = _write(tmp_path, n=21)
    v = Vocab(voc)
    ds = Dataset(src, tgt, v, v, max_len=50, batch_or_token="batch", data_leak_ratio=0.5)
    got = [d for d in ds.batcher(8, buffer_size=1000, shuffle=False, train=True)]
    assert [len(d['index']) for d in got] == [8, 8, 5]      # 5 >= 8*0.5 is kept
    ds = Dataset(src, tgt, v, v, max_len=50, batch_or_token="batch", data_leak_ratio=0.9)
    got = [d for d in ds.batcher(8, buffer_size=1000, shuffle=False, train=True)]
    assert [len(d['index']) for d in got] == [8, 8] and len(ds.leak_buffer) == 5
    # ids: eos appended after truncation, unknown words -> <unk>=1
    ids = v.to_id("w1 w25 w3".split())
    assert ids[-1] == 2 and ids[1] == 1


# ---- round 5: the row pinned by the reference's OWN code ---------------------------------------
# tests/golden/make_data_golden.py takes batch_indexer / token_indexer (util.py:17-65) and Dataset (data.py:11-117) out
# of the reference's syntax trees and runs them unchanged (the modules themselves import TensorFlow); reference_data.json
# holds their inputs and outputs.
import json
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data.json")


def test_indexers_equal_the_reference_functions():
    fx = json.load(open(GOLD))
    assert len(fx["batch_indexer"]) >= 8 and len(fx["token_indexer"]) >= 60
    for c in fx["batch_indexer"]:
        assert batch_indexer(c["datasize"], c["batch_size"]) == c["batches"], c
    for c in fx["token_indexer"]:
        assert token_indexer([tuple(l) for l in c["lens"]], c["token_size"]) == c["batches"], (c["lens"][:5], c["token_size"])


def test_dataset_batcher_equals_the_reference_class(tmp_path):
    """Two passes per case (the leak buffer of one pass opens the next, data.py:98-99), token and sentence batching,
    shuffled (numpy's global stream, seeded as the generator seeded it) and not, train / eval tail rule."""
    fx = json.load(open(GOLD))
    assert len(fx["dataset"]) >= 14
    for ci, c in enumerate(fx["dataset"]):
        sp, tp, vp = (str(tmp_path / ("%s%d.txt" % (x, ci))) for x in "stv")
        open(sp, "w").write("\n".join(c["src_lines"]) + "\n")
        open(tp, "w").write("\n".join(c["tgt_lines"]) + "\n")
        open(vp, "w").write("\n".join(c["vocab_lines"]) + "\n")
        v = Vocab(vp)
        ds = Dataset(sp, tp, v, v, max_len=c["max_len"], batch_or_token=c["mode"], data_leak_ratio=c["data_leak_ratio"])
        for epoch, want in enumerate(c["epochs"]):
            np.random.seed(c["seed_base"] + epoch)
            got = list(ds.batcher(c["size"], buffer_size=c["buffer_size"], shuffle=c["shuffle"], train=c["train"]))
            assert len(got) == len(want["batches"]), (ci, epoch, len(got), len(want["batches"]))
            for g, w in zip(got, want["batches"]):
                assert [int(i) for i in g["index"]] == w["index"], (ci, epoch)
                assert g["src"].dtype == np.int32 and g["src"].tolist() == w["src"] and g["tgt"].tolist() == w["tgt"], (ci, epoch)
            assert [int(s[0]) for s in ds.leak_buffer] == want["leak_index"], (ci, epoch)


# ---- and against an independently written statement of the rule, over random inputs (hypothesis) ----------------------
from hypothesis import given, settings, strategies as hst


def _brute_token_batches(lens, token_size):
    """util.py:30-65 stated from its behaviour, not its loop: scan the samples in order; a batch grows while, WITH the
    next sample included, count x (per-side running maximum) stays below token_size on every side; the sample that
    reaches the budget is NOT included (it opens the next batch) unless the batch is empty, in which case it is a batch
    of its own.  Whatever is left when the scan ends without reaching the budget forms the tail batch."""
    out, cur = [], []
    i = 0
    while i < len(lens):
        trial = cur + [i]
        width = len(lens[0])
        over = any(len(trial) * max(lens[j][s] for j in trial) >= token_size for s in range(width))
        if not over:
            cur = trial
            i += 1
        elif not cur:
            out.append([i])
            i += 1
        else:
            out.append(cur)
            cur = []
    if cur:
        out.append(cur)
    return out


@settings(max_examples=300, deadline=None)
@given(hst.lists(hst.tuples(hst.integers(1, 80), hst.integers(1, 80)), min_size=0, max_size=120), hst.integers(1, 2000))
def test_token_indexer_properties(lens, token_size):
    got = token_indexer(lens, token_size)
    assert got == _brute_token_batches(lens, token_size)
    flat = [i for b in got for i in b]
    assert flat == list(range(len(lens)))                                   # every sample once, in order
    for bi, b in enumerate(got):
        if len(b) > 1 or bi == len(got) - 1:
            continue
        # a closed one-sample batch: either the sample alone reaches the budget, or the next one would have
        alone = any(l >= token_size for l in lens[b[0]])
        nxt = b[0] + 1
        assert alone or (nxt < len(lens) and any(2 * max(lens[b[0]][s], lens[nxt][s]) >= token_size for s in range(2)))
    for b in got[:-1]:
        if len(b) > 1:      # closed BEFORE the budget: what it holds is strictly below it on both sides
            assert all(len(b) * max(lens[j][s] for j in b) < token_size for s in range(2))


@settings(max_examples=200, deadline=None)
@given(hst.integers(0, 300), hst.integers(1, 70))
def test_batch_indexer_properties(n, size):
    got = batch_indexer(n, size)
    assert [i for b in got for i in b] == list(range(n))
    assert all(len(b) == size for b in got[:-1]) and (not got or 1 <= len(got[-1]) <= size)
