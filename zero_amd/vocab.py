# coding: utf-8
"""Vocabulary with the reference's reserved ids (vocab.py:10-81): <pad>=0, <unk>=1,
<eos>=2 inserted first, one token per line in the vocabulary file, ``to_id`` appends
<eos> by default (vocab.py:69-73)."""


class Vocab(object):
    pad_sym, unk_sym, eos_sym = "<pad>", "<unk>", "<eos>"

    def __init__(self, vocab_file=None):
        self.word2id, self.id2word = {}, {}
        for tok in (self.pad_sym, self.unk_sym, self.eos_sym):
            self.insert(tok)
        if vocab_file is not None:
            with open(vocab_file, "r") as reader:
                for line in reader:
                    self.insert(line.strip())

    def insert(self, token):
        if token not in self.word2id:
            idx = len(self.word2id)
            self.word2id[token] = idx
            self.id2word[idx] = token

    def size(self):
        return len(self.word2id)

    def get_id(self, token):
        return self.word2id.get(token, self.word2id[self.unk_sym])

    def get_token(self, idx):
        return self.id2word.get(idx, self.unk_sym)

    def to_id(self, tokens, append_eos=True):
        toks = list(tokens) + ([self.eos_sym] if append_eos else [])
        return [self.get_id(t) for t in toks]

    def to_tokens(self, ids):
        return [self.get_token(i) for i in ids]

    def eos(self):
        return self.get_id(self.eos_sym)

    def pad(self):
        return self.get_id(self.pad_sym)
