# coding: utf-8
"""Vocabulary with the reference's reserved ids (vocab.py:10-81): <pad>=0, <unk>=1,
<eos>=2 come first, the vocabulary file holds one token per line, ``to_id`` appends
<eos> by default (vocab.py:69-73).

Also the vocabulary PREPARATION tool of the reference (vocab.py:84-103): count the tokens of a corpus, order
them by falling frequency (ties keep first-seen order, the three reserved symbols stay in front) and write the
first ``size`` entries::

    python -m zero_amd.vocab [--size N] corpus.txt vocab.txt

Outputs are held against the reference's own module on seeded corpora (tests/golden/reference_vocab.json).
"""
import argparse
import collections

RESERVED = ("<pad>", "<unk>", "<eos>")


class Vocab(object):
    pad_sym, unk_sym, eos_sym = RESERVED

    def __init__(self, vocab_file=None):
        self._tokens = []                            # id -> token
        self._index = {}                             # token -> id
        self._seen = collections.Counter()           # token -> number of insert() calls (first-seen order)
        self._renumber(())
        if vocab_file is not None:
            with open(vocab_file, "r") as reader:
                for line in reader:
                    self.insert(line.strip())

    # ---- construction
    def insert(self, token):
        """A new token takes the next id; every call counts one occurrence (vocab.py:26-33)."""
        if token not in self._index:
            self._index[token] = len(self._tokens)
            self._tokens.append(token)
        self._seen[token] += 1

    def _renumber(self, ordered):
        """Ids from scratch: the reserved symbols, then ``ordered``; each listing counts as an occurrence, as a
        re-insertion does in the reference (its counts grow by one per sort)."""
        self._tokens, self._index = [], {}
        for token in RESERVED + tuple(ordered):
            self.insert(token)

    def sort_vocab(self):
        """vocab.py:53-61: by falling count; Python's stable sort keeps first-seen order among equal counts."""
        ranked = sorted(self._seen, key=lambda token: -self._seen[token])
        self._renumber(ranked)

    def save_vocab(self, vocab_file, size=1e6):
        keep = self._tokens[:min(len(self._tokens), int(size))]
        with open(vocab_file, "w") as writer:
            writer.writelines(token + "\n" for token in keep)

    # ---- lookup
    def size(self):
        return len(self._tokens)

    def get_id(self, token):
        return self._index.get(token, self._index[self.unk_sym])

    def get_token(self, idx):
        return self._tokens[idx] if 0 <= idx < len(self._tokens) else self.unk_sym

    def to_id(self, tokens, append_eos=True):
        ids = [self.get_id(token) for token in tokens]
        return ids + [self.eos()] if append_eos else ids

    def to_tokens(self, ids):
        return [self.get_token(i) for i in ids]

    def eos(self):
        return self._index[self.eos_sym]

    def pad(self):
        return self._index[self.pad_sym]


def build_vocab(corpus_file, vocab_file, size=1e6):
    """vocab.py:84-103: whitespace tokens of every line, sorted by count, first ``size`` entries written."""
    vocab = Vocab()
    with open(corpus_file, "r") as reader:
        for line in reader:
            for token in line.split():
                vocab.insert(token)
    vocab.sort_vocab()
    vocab.save_vocab(vocab_file, size)
    return vocab


def main(argv=None):
    parser = argparse.ArgumentParser("Vocabulary Preparison")
    parser.add_argument("--size", type=int, default=1e6, help="maximum vocabulary size")
    parser.add_argument("input", type=str, help="the input file path")
    parser.add_argument("output", type=str, help="the output file name")
    args = parser.parse_args(argv)
    vocab = build_vocab(args.input, args.output, args.size)
    print("Loading {} tokens from {}".format(vocab.size(), args.input))


if __name__ == "__main__":
    main()
