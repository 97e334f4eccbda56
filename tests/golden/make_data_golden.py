# coding: utf-8
"""Golden vectors for the data-path row of SURVEY.md 8(f)-1 (run in the BUILD container only).

``/root/reference/utils/util.py`` and ``/root/reference/data.py`` cannot be IMPORTED here: util.py imports TensorFlow
at module level (util.py:12) and data.py imports util.  The functions on the data path, however, are plain Python:
``batch_indexer`` / ``token_indexer`` (util.py:17-65) use numpy only and ``Dataset`` (data.py:11-117) uses them.  So
this script reads the two reference files, takes exactly those three definitions out of their syntax trees (``ast``),
compiles them UNCHANGED and runs them -- the reference's own code, not a restatement -- on seeded inputs; the
reference's ``vocab.py`` is imported normally.  Nothing of the reference's text is stored: the fixture
``reference_data.json`` holds inputs (length lists, corpora, vocabulary files, arguments) and outputs (index lists,
batch matrices, leak buffers).  tests/test_data.py replays the inputs through zero_amd/data.py.
"""
import ast
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _definitions(path, names, namespace):
    tree = ast.parse(open(path).read(), filename=path)
    picked = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in picked) == sorted(names), (path, [n.name for n in picked])
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), namespace)
    return namespace


def main():
    sys.path.insert(0, REF)
    import vocab as ref_vocab                                      # reference code, imported
    ns = {"np": np}
    _definitions(os.path.join(REF, "utils", "util.py"), ["batch_indexer", "token_indexer"], ns)
    _definitions(os.path.join(REF, "data.py"), ["Dataset"], ns)
    batch_indexer, token_indexer, Dataset = ns["batch_indexer"], ns["token_indexer"], ns["Dataset"]
    rnd = random.Random(20260928)
    out = {"batch_indexer": [], "token_indexer": [], "dataset": []}
    for n, b in [(0, 3), (1, 1), (7, 3), (6, 3), (2, 5), (100, 7), (64, 64), (65, 64)]:
        out["batch_indexer"].append({"datasize": n, "batch_size": b, "batches": batch_indexer(n, b)})
    for case in range(60):
        n = rnd.choice([1, 2, 3, 10, 50, 200, 400])
        width = rnd.choice([1, 2, 2, 2, 3])
        hi = rnd.choice([4, 12, 64, 120])
        if case % 5 == 0:       # sorted by the longer side, as the batcher hands them over
            lens = sorted(([rnd.randint(1, hi) for _ in range(width)] for _ in range(n)), key=max)
        elif case % 5 == 1:     # constant lengths: the 63-not-64 rule
            lens = [[hi] * width for _ in range(n)]
        else:
            lens = [[rnd.randint(1, hi) for _ in range(width)] for _ in range(n)]
        if case % 7 == 3 and n > 2:      # an instance that alone exceeds the budget
            lens[rnd.randrange(n)] = [hi * 50] * width
        size = rnd.choice([hi, 2 * hi, 10 * hi, 64 * hi, 100 * hi + 1, hi * hi])
        out["token_indexer"].append({"lens": lens, "token_size": size, "batches": token_indexer(lens, size)})
    words = ["w%d" % i for i in range(40)]
    for case in range(14):
        nlines = [57, 21, 21, 200, 5, 130, 64, 300, 33, 90, 1, 150, 77, 260][case]
        mode = "token" if case % 2 else "batch"
        max_len = [8, 50, 50, 12, 3, 20, 100, 9, 6, 30, 5, 16, 10, 25][case]
        size = ([8, 8, 8, 16, 2, 10, 64, 32, 4, 7, 3, 20, 5, 12][case] if mode == "batch"
                else [0, 60, 0, 150, 0, 90, 0, 250, 0, 40, 0, 400, 0, 33][case])
        leak = [0.5, 0.5, 0.9, 0.5, 0.0, 0.75, 0.5, 0.3, 1.0, 0.5, 0.5, 0.6, 0.5, 0.99][case]
        buffer_size = [20, 1000, 1000, 50, 2, 64, 1000, 100, 10, 1000, 1000, 37, 1000, 128][case]
        shuffle = case % 3 == 0
        train = case % 4 != 1
        src_lines, tgt_lines = [], []
        for _ in range(nlines):
            src_lines.append(" ".join(rnd.choices(words, k=rnd.randint(0 if case == 3 else 1, 14))))
            tgt_lines.append(" ".join(rnd.choices(words, k=rnd.randint(1, 14))) + ("  " if rnd.random() < 0.1 else ""))
        vocab_lines = words[:25]
        with tempfile.TemporaryDirectory() as d:
            sp, tp, vp = (os.path.join(d, x) for x in ("s.txt", "t.txt", "v.txt"))
            open(sp, "w").write("\n".join(src_lines) + "\n")
            open(tp, "w").write("\n".join(tgt_lines) + "\n")
            open(vp, "w").write("\n".join(vocab_lines) + "\n")
            v = ref_vocab.Vocab(vp)
            ds = Dataset(sp, tp, v, v, max_len=max_len, batch_or_token=mode, data_leak_ratio=leak)
            epochs = []
            for epoch in range(2):          # the leak buffer of one pass opens the next (data.py:98-99)
                np.random.seed(1000 + case * 10 + epoch)
                batches = []
                for data in ds.batcher(size, buffer_size=buffer_size, shuffle=shuffle, train=train):
                    batches.append({"index": [int(i) for i in data["index"]], "src": data["src"].tolist(),
                                    "tgt": data["tgt"].tolist()})
                epochs.append({"batches": batches, "leak_index": [int(s[0]) for s in ds.leak_buffer]})
        out["dataset"].append({"src_lines": src_lines, "tgt_lines": tgt_lines, "vocab_lines": vocab_lines, "mode": mode,
                               "max_len": max_len, "size": size, "data_leak_ratio": leak, "buffer_size": buffer_size,
                               "shuffle": shuffle, "train": train, "seed_base": 1000 + case * 10, "epochs": epochs})
    json.dump(out, open(os.path.join(HERE, "reference_data.json"), "w"), separators=(",", ":"))
    print("wrote %d + %d + %d cases" % (len(out["batch_indexer"]), len(out["token_indexer"]), len(out["dataset"])))


if __name__ == "__main__":
    main()
