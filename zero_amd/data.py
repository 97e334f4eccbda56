# coding: utf-8
"""Host data path feeding the step: bitext -> ids -> length-sorted token/sentence batches.

Next-row component, restating the reference's observable batching
semantics: ``utils/util.py:17-65`` (``batch_indexer``, ``token_indexer``) and
``data.py:11-117`` (``Dataset``: truncation to ``max_len`` BEFORE appending <eos>
(data.py:42-45, vocab.py:69-73), zero-padded int32 matrices (data.py:47-65), buffers of
``buffer_size`` sentences sorted by max(src_len, tgt_len), optional shuffling of the batch
order, and the "leak" rule that carries tail batches smaller than
``size * data_leak_ratio`` over to the next buffer (data.py:98-117)).

The reference modules cannot be IMPORTED here (utils/util.py imports TensorFlow at module level), but the three
definitions on this path are plain Python: tests/golden/make_data_golden.py lifts ``batch_indexer`` / ``token_indexer``
/ ``Dataset`` out of the reference's syntax trees, runs them unchanged on seeded inputs and stores inputs + outputs
(tests/golden/reference_data.json); tests/test_data.py replays them through this module -- the row is pinned by the
reference's own code (round 5) -- beside hand-derived known answers and hypothesis properties against an independently
written statement of the batching rule.
"""

import numpy as np


def batch_indexer(datasize, batch_size):
    """util.py:17-27: consecutive chunks of ``batch_size`` indices (+ the remainder)."""
    idx = list(range(datasize))
    out = [idx[i * batch_size:(i + 1) * batch_size] for i in range(datasize // batch_size)]
    if datasize % batch_size > 0:
        out.append(idx[-(datasize % batch_size):])
    return out


def token_indexer(dataset, token_size):
    """util.py:30-65.  ``dataset`` = [(len_1, ..., len_n)] per sample.  A batch is closed
    BEFORE the sample that would make count * max_len reach ``token_size`` on any side
    (so 64-token sentences with token_size=4096 give 63-sentence batches); a single sample
    that alone reaches the budget becomes a 1-sample batch."""
    n = len(dataset)
    if n == 0:
        return []
    width = len(dataset[0])
    out = []
    maxes = [0.0] * width
    count = 0
    i = 0
    while i < n:
        maxes = [max(m, l) for m, l in zip(maxes, dataset[i])]
        count += 1
        if any(count * l >= token_size for l in maxes):
            if count > 1:
                out.append(list(range(i - count + 1, i)))
                i -= 1
            else:
                out.append([i])
            count = 0
            maxes = [0.0] * width
        i += 1
    done = sum(len(b) for b in out)
    if done != n:
        out.append(list(range(done, n)))
    return out


class Dataset(object):
    def __init__(self, src_file, tgt_file, src_vocab, tgt_vocab, max_len=100, batch_or_token='batch',
                 data_leak_ratio=0.5):
        self.source, self.target = src_file, tgt_file
        self.src_vocab, self.tgt_vocab = src_vocab, tgt_vocab
        self.max_len = max_len
        self.batch_or_token = batch_or_token
        self.data_leak_ratio = data_leak_ratio
        self.leak_buffer = []

    def load_data(self):
        with open(self.source, 'r') as sr, open(self.target, 'r') as tr:
            while True:
                s, t = sr.readline(), tr.readline()
                if s == "" or t == "":
                    break
                s, t = s.strip(), t.strip()
                if s == "" or t == "":
                    continue
                yield (self.src_vocab.to_id(s.split()[:self.max_len]),
                       self.tgt_vocab.to_id(t.split()[:self.max_len]))

    def to_matrix(self, batch):
        src_len = min(self.max_len, max(len(b[1]) for b in batch))
        tgt_len = min(self.max_len, max(len(b[2]) for b in batch))
        s = np.zeros([len(batch), src_len], dtype=np.int32)
        t = np.zeros([len(batch), tgt_len], dtype=np.int32)
        for i, (_, si, ti) in enumerate(batch):
            s[i, :min(src_len, len(si))] = si[:src_len]
            t[i, :min(tgt_len, len(ti))] = ti[:tgt_len]
        return [b[0] for b in batch], s, t

    def _handle_buffer(self, buf, size, shuffle):
        buf = sorted(buf, key=lambda x: max(len(x[1]), len(x[2])))
        if self.batch_or_token == 'batch':
            index = batch_indexer(len(buf), size)
        else:
            index = token_indexer([[len(b[1]), len(b[2])] for b in buf], size)
        order = list(range(len(index)))
        if shuffle:
            np.random.shuffle(order)
        for o in order:
            batch = [buf[i] for i in index[o]]
            x, s, t = self.to_matrix(batch)
            yield {'src': s, 'tgt': t, 'index': x, 'raw': batch}

    def _size_of(self, data):
        if self.batch_or_token == 'batch':
            return len(data['raw'])
        return max(np.sum(data['tgt'] > 0), np.sum(data['src'] > 0))

    def batcher(self, size, buffer_size=1000, shuffle=True, train=True):
        buf = self.leak_buffer
        self.leak_buffer = []
        for i, (s, t) in enumerate(self.load_data()):
            buf.append((i, s, t))
            if len(buf) >= buffer_size:
                for data in self._handle_buffer(buf, size, shuffle):
                    if self._size_of(data) < size * self.data_leak_ratio:
                        self.leak_buffer += data['raw']
                    else:
                        yield data
                buf = self.leak_buffer
                self.leak_buffer = []
        if buf:
            for data in self._handle_buffer(buf, size, shuffle):
                if train and self._size_of(data) < size * self.data_leak_ratio:
                    self.leak_buffer += data['raw']
                else:
                    yield data


def features_of(batch):
    """The {"source", "target"} dict train_fn / score_fn take (main.py:286-294)."""
    return {"source": batch['src'], "target": batch['tgt']}
