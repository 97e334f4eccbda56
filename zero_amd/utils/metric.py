# coding: utf-8
"""Corpus BLEU / OTEM / UTEM on tokenised text (the reference's utils/metric.py; evaluation row of
SURVEY.md §8(f)-4).  Host code: token lists in, a float out.

Restated around one shared corpus scan (`_Corpus`) instead of three copies of the loop; the
arithmetic follows utils/metric.py line by line where it is observable:
  * n-gram tables per sentence, n = 1..N (metric.py:52-59);
  * reference length per sentence: closest (ties -> shorter) or shortest (metric.py:70-88);
  * BLEU clipped matches = max over references of min(ref, cand) (metric.py:263-271), brevity
    penalty ``exp(1 - r/c)`` when ``c <= r`` (metric.py:286-287);
  * OTEM over-translation counts = min over references of the surplus (metric.py:118-140), length
    factor ``exp(1 - r/c)`` when ``c >= r`` (metric.py:158-159);
  * UTEM missing counts per order = min over references, totals = max over references
    (metric.py:190-218), length factor ``exp(1 - c/r)`` when ``c <= r`` (metric.py:234-235);
  * geometric mean with a guarded log: a non-positive precision contributes -9999999999
    (metric.py:91-97); optional +1 smoothing of orders > 1 (metric.py:151-153).
Pinned against the reference module itself (importable without TensorFlow):
tests/golden/reference_host.json.
"""

import math
import os
import sys
from collections import Counter


def _ngrams(tokens, n):
    table = Counter()
    for order in range(1, n + 1):
        for i in range(len(tokens) - order + 1):
            table[tuple(tokens[i:i + order])] += 1
    return table


def _ref_length(ref_lengths, cand_length, closest=True):
    if not closest:
        return min(ref_lengths)
    best, gap = 9999, 9999
    for r in ref_lengths:
        d = abs(r - cand_length)
        if d < gap or (d == gap and r < best):
            best, gap = r, d
    return best


def _safe_log(x):
    if x <= 0:
        print("WARNING, a non-positive number is processed by log", file=sys.stderr)
        return -9999999999
    return math.log(x)


def _finish(num, den, n, smooth, weights, length_factor):
    ratios = {}
    for order in range(1, n + 1):
        if order in den:
            a, b = num.get(order, 0), den[order]
            if smooth and order > 1:
                a, b = a + 1, b + 1
            ratios[order] = a * 1.0 / b
    if weights is None:
        weights = [1.0 / n] * n
    assert len(weights) == n, \
        'ERROR: the length of weights ({}) should be equal to n ({})'.format(len(weights), n)
    return length_factor * math.exp(sum(_safe_log(ratios.get(i + 1, 0)) * weights[i] for i in range(n)))


def _lengths(cand, refs, bp):
    len_c = len_r = 0
    for c, rs in zip(cand, refs):
        len_c += len(c)
        len_r += _ref_length([len(r) for r in rs], len(c), closest=(bp == 'closest'))
    return len_c, len_r


def bleu(cand, refs, bp='closest', smooth=False, n=4, weights=None):
    """BLEU-n of a corpus: ``cand`` list of token lists, ``refs`` list (per sentence) of
    reference token lists.  LARGER is better.  metric.py:243-297."""
    total, match = Counter(), Counter()
    for c, rs in zip(cand, refs):
        cg = _ngrams(c, n)
        tables = [_ngrams(r, n) for r in rs]
        for gram, cnt in cg.items():
            total[len(gram)] += cnt if tables else 0
            match[len(gram)] += max([min(t.get(gram, 0), cnt) for t in tables] or [0])
    len_c, len_r = _lengths(cand, refs, bp)
    if len_r == 0:
        return 0.
    factor = math.exp(1. - len_r * 1. / len_c) if len_c <= len_r else 1.
    return _finish(match, total, n, smooth, weights, factor)


def otem(cand, refs, bp='closest', smooth=False, n=2, weights=None):
    """Over-translation metric, LOWER is better.  metric.py:100-168."""
    total, over = Counter(), Counter()
    for c, rs in zip(cand, refs):
        cg = _ngrams(c, n)
        tables = [_ngrams(r, n) for r in rs]
        for gram, cnt in cg.items():
            total[len(gram)] += cnt if tables else 0
            surplus = []
            for t in tables:
                if gram not in t:
                    s = cnt - 1 if cnt > 1 else 0
                else:
                    s = cnt - t[gram] if cnt > t[gram] else 0
                if s > 0:
                    surplus.append(s)
            over[len(gram)] += min(surplus) if surplus else 0
    len_c, len_r = _lengths(cand, refs, bp)
    if len_r == 0:
        return 0.
    factor = math.exp(1. - len_r * 1. / len_c) if len_c >= len_r else 1.
    return _finish(over, total, n, smooth, weights, factor)


def utem(cand, refs, bp='closest', smooth=False, n=4, weights=None):
    """Under-translation metric, LOWER is better.  metric.py:171-240."""
    total, miss = Counter(), Counter()
    for c, rs in zip(cand, refs):
        cg = _ngrams(c, n)
        per_ref_total, per_ref_miss = {}, {}
        for r in rs:
            t_ref, m_ref = Counter(), Counter()
            for gram, cnt in _ngrams(r, n).items():
                t_ref[len(gram)] += cnt
                have = cg.get(gram, 0)
                m_ref[len(gram)] += cnt - have if cnt > have else 0
            for order in t_ref:
                per_ref_total.setdefault(order, []).append(t_ref[order])
                per_ref_miss.setdefault(order, []).append(m_ref[order])
        for order in per_ref_total:
            miss[order] += min(per_ref_miss[order])
            total[order] += max(per_ref_total[order])
    len_c, len_r = _lengths(cand, refs, bp)
    if len_r == 0:
        return 0.
    factor = math.exp(1. - len_c * 1. / len_r) if len_c <= len_r else 1.
    return _finish(miss, total, n, smooth, weights, factor)


def get_refs(ref):
    """Reference files following the multi-bleu convention: ``ref`` itself if it exists, else
    ``ref0, ref1, ...`` (metric.py:16-36); None when neither exists."""
    if os.path.exists(ref):
        return [ref]
    found = []
    while os.path.exists(ref + "%d" % len(found)):
        found.append(ref + "%d" % len(found))
    return found or None


def read_tokens(path, lc=False):
    with open(path, "r") as f:
        return [(line.strip().lower() if lc else line.strip()).split() for line in f.readlines()]


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="OTEM-2 / UTEM-4 / BLEU-4 on multiple references")
    ap.add_argument('-lc', action='store_true')
    ap.add_argument('-bp', default='closest', choices=['shortest', 'closest'])
    ap.add_argument('candidate')
    ap.add_argument('reference')
    a = ap.parse_args()
    files = get_refs(a.reference)
    if files is None:
        print('Error: could not find proper reference file ', a.reference + "0", file=sys.stderr)
        sys.exit(1)
    cands = read_tokens(a.candidate, a.lc)
    per_ref = [read_tokens(f, a.lc) for f in files]
    assert len(cands) == len(per_ref[0]), 'ERROR: the length of candidate and reference must be the same.'
    grouped = list(zip(*per_ref))
    print('OTEM-2/UTEM-4/BLEU-4: {}/{}/{}'.format(otem(cands, grouped, bp=a.bp, n=2), utem(cands, grouped, bp=a.bp, n=4),
                                                  bleu(cands, grouped, bp=a.bp, n=4)))
