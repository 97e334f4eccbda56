# coding: utf-8
"""The fitted EOS row of the decode fixture's softmax table (tests/fullsize.py beam_params; run in the BUILD container).

A trained translation model stops when it has covered its source; a random one never does (or does at random).  The
decode fixture gets the stopping rule by FITTING one row of the softmax table, which is what training would do to it:

  1. the fixture's weight set with a zero EOS row decodes 96 training sentences (beam_sources(96, seed=99): other
     sentences than the fixture's) greedily with the fp32 oracle; no EOS can be emitted, so every sentence runs to its
     length cap and every (sentence, step) yields the decoder feature f[b, t] (recovered from the logits by the
     pseudo-inverse of the other 31997 rows);
  2. ridge regression  e = argmin sum (e . f[b, t] - y[b, t])^2 + lambda |e|^2  with
     y[b, t] = median top logit + kappa (t - (Ls_b - 1)),  Ls_b = source length incl. its eos:  the EOS logit overtakes the
     best other candidate around the step at which the hypothesis is as long as the source, and grows by kappa per step.

The information is in the feature: the decoder input carries the timing signal of t (func.py:341-369), and the
near-uniform cross attention averages the ENCODER's timing signal over the source positions -- its low-frequency
channels are proportional to the source length.  Residual of the fit: see the printed line (about +-3.4 steps).

Writes tests/golden/aan_base_beam_eos_row.npy (512 x fp32).  ~2 minutes on 8 cores.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_torch as rt  # noqa: E402
from tests.fullsize import (beam_hp, beam_params, beam_sources, BEAM_EOS_KAPPA, BEAM_EOS_RIDGE, BEAM_EOS_ROW)  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    hp = beam_hp()
    hp.beam_size = 1
    model = hp.model_name
    Pn = beam_params(hp, model, eos_row=None)
    Es = Pn["softmax_embedding"]
    H = Es.shape[1]
    pinv = np.linalg.pinv(Es[3:].astype(np.float64))
    P = rt.to_torch(Pn)
    enc, dec = rt.infer_fn(hp, P, model)
    train = beam_sources(96, seed=99)
    X, T, L, TOP = [], [], [], []
    cur = {}

    def dec_rec(target, state, t):
        lg, st = dec(target, state, t)
        a = lg.numpy().astype(np.float64)
        X.append(a[:, 3:] @ pinv.T)
        T.append(np.full(a.shape[0], t))
        L.append(cur["len"].copy())
        TOP.append(a.max(1))
        return lg, st
    t0 = time.time()
    for i in range(0, train.shape[0], 32):
        src = train[i:i + 32]
        cur["len"] = (src != 0).sum(1)
        rt.beam_search({"source": torch.tensor(src)}, enc, dec_rec, hp)
    X, T, L, TOP = (np.concatenate(v) for v in (X, T, L, TOP))
    keep = T <= L + 25                     # the steps far past the source length say nothing about WHEN to stop
    X, T, L, TOP = X[keep], T[keep], L[keep], TOP[keep]
    y = np.median(TOP) + BEAM_EOS_KAPPA * (T - (L - 1))
    e = np.linalg.solve(X.T @ X + BEAM_EOS_RIDGE * np.eye(H), X.T @ y)
    res = X @ e - y
    print("features of %d (sentence, step) pairs in %.0f s; |e| = %.2f (other rows: %.2f), residual std %.3f logits = +-%.1f "
          "steps at kappa = %.2f, median top logit %.2f" % (len(y), time.time() - t0, np.linalg.norm(e),
                                                           float(np.linalg.norm(Es[3:], axis=1).mean()), res.std(),
                                                           res.std() / BEAM_EOS_KAPPA, BEAM_EOS_KAPPA, np.median(TOP)))
    np.save(os.path.join(HERE, BEAM_EOS_ROW), e.astype(np.float32))


if __name__ == "__main__":
    main()
