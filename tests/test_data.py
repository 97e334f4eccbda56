"""Host data path: batching semantics of util.py:17-65 / data.py:11-117,
pinned by hand-derived known answers (the reference module needs TensorFlow to import)."""
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
