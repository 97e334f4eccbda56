# coding: utf-8
"""Golden vectors for the hypothesis cut and two small host helpers (run in the BUILD
container only).  ``evalu.py`` and ``utils/util.py`` import TensorFlow at module level, but ``decode_target_token`` /
``decode_hypothesis`` (evalu.py:14-46) and ``closing_dropout`` (util.py:106-114) are plain Python: they are taken out of
the reference's syntax trees (``ast``), compiled UNCHANGED and run on seeded inputs, with the reference's own ``vocab.py``
imported normally.  reference_evalu.json holds inputs and outputs only."""
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


class _Params(object):
    """What closing_dropout / decode_hypothesis touch of tf.contrib's HParams: attributes + values()."""

    def __init__(self, d):
        self.__dict__.update(d)

    def values(self):
        return {k: v for k, v in self.__dict__.items() if k != "tgt_vocab"}


def main():
    sys.path.insert(0, REF)
    import vocab as ref_vocab
    ns = {"np": np}
    _definitions(os.path.join(REF, "evalu.py"), ["decode_target_token", "decode_hypothesis"], ns)
    _definitions(os.path.join(REF, "utils", "util.py"), ["closing_dropout"], ns)
    rnd = random.Random(20260928)
    words = ["w%d" % i for i in range(30)]
    out = {"vocab_lines": words, "hypothesis": [], "closing_dropout": []}
    with tempfile.TemporaryDirectory() as d:
        vp = os.path.join(d, "v.txt")
        open(vp, "w").write("\n".join(words) + "\n")
        v = ref_vocab.Vocab(vp)
        for case in range(12):
            towers = rnd.randint(1, 3)
            seqs, scores = [], []
            for _ in range(towers):
                B, K, L = rnd.randint(1, 5), rnd.choice([1, 2, 4]), rnd.randint(1, 9)
                s = [[[rnd.choice([0, 1, 2, 2] + list(range(3, v.size() + 3))) for _ in range(L)] for _ in range(K)] for _ in range(B)]
                seqs.append(s)
                scores.append([[round(rnd.uniform(-9, 0), 4) for _ in range(K)] for _ in range(B)])
            mask = None if case % 3 == 0 else [float(rnd.random() < 0.7) for _ in range(towers)]
            hyp, marks = ns["decode_hypothesis"](seqs, scores, _Params({"tgt_vocab": v}), mask=mask)
            out["hypothesis"].append({"seqs": seqs, "scores": scores, "mask": mask, "hypoes": hyp, "marks": marks})
    for case in range(4):
        hp = {"dropout": 0.1, "relu_dropout": 0.2, "attention_dropout": 0.3, "residual_dropout": 0.05, "label_smooth": 0.1,
              "label_smoothing_rate": 0.2, "hidden_size": 512, "dropout_keep": 0.9, "name": "x", "l0_dropout_scale": 3}
        keys = list(hp)
        rnd.shuffle(keys)
        hp = {k: hp[k] for k in keys[:6 + case]}
        p = ns["closing_dropout"](_Params(dict(hp)))
        out["closing_dropout"].append({"before": hp, "after": p.values()})
    json.dump(out, open(os.path.join(HERE, "reference_evalu.json"), "w"), separators=(",", ":"))
    print("wrote", len(out["hypothesis"]), "+", len(out["closing_dropout"]), "cases")


if __name__ == "__main__":
    main()
