# coding: utf-8
"""Golden vectors for the vocabulary row of SURVEY.md §8(f) (run in the BUILD container only): the reference's
``vocab.py`` is IMPORTED from /root/reference (never copied) and run on seeded corpora; inputs and outputs go to
``reference_vocab.json``."""
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, "/root/reference")
    import vocab as ref_vocab                      # reference code, imported
    rnd = random.Random(20260928)
    cases = []
    for case in range(6):
        nwords = [5, 40, 300, 40, 1, 25][case]
        words = ["tok%d" % i for i in range(nwords)] + (["<unk>", "<eos>"] if case == 3 else [])
        weights = [1.0 / (1 + i) for i in range(len(words))]
        lines = []
        for _ in range([3, 60, 400, 60, 2, 0][case]):
            n = rnd.randint(0, 12)
            lines.append(" ".join(rnd.choices(words, weights=weights, k=n)) + ("  " if rnd.random() < 0.2 else ""))
        size = [1e6, 20, 100, 1e6, 2, 1e6][case]
        with tempfile.TemporaryDirectory() as d:
            src, out = os.path.join(d, "corpus.txt"), os.path.join(d, "vocab.txt")
            open(src, "w").write("\n".join(lines) + ("\n" if lines else ""))
            v = ref_vocab.Vocab()
            with open(src, "r") as reader:
                for line in reader:
                    for token in line.strip().split():
                        v.insert(token)
            v.sort_vocab()
            v.save_vocab(out, size)
            written = open(out).read()
            loaded = ref_vocab.Vocab(out)
        probe = [rnd.choice(words + ["never-seen", "<pad>"]) for _ in range(12)]
        ids = [rnd.randint(0, loaded.size() + 3) for _ in range(12)]
        cases.append({"corpus": lines, "size": size, "vocab_file": written, "loaded_size": loaded.size(),
                      "probe": probe, "to_id": loaded.to_id(probe), "to_id_no_eos": loaded.to_id(probe, append_eos=False),
                      "ids": ids, "to_tokens": loaded.to_tokens(ids),
                      "eos": loaded.eos(), "pad": loaded.pad(), "sorted_size": v.size()})
    json.dump({"cases": cases}, open(os.path.join(HERE, "reference_vocab.json"), "w"), indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
