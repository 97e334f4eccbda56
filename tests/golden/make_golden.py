# coding: utf-8
"""Regenerates the committed golden fixtures (run in the BUILD container only).

1. ``reference_scalars.json`` -- values produced by IMPORTING the reference's own TF-free
   modules from /root/reference (lrs/noamlr.py, vocab.py).  These are the only reference
   code paths that can execute here (no TensorFlow); they pin the host-side scalars.
2. ``tiny_<model>.npz`` -- inputs and expected outputs of the CPU oracle (fp64 torch
   restatement, cross-checked against the numpy restatement) for the four registered
   models on a tiny config: parameters by reference variable name, ids, loss, per-sentence
   loss, gradient norms, beam-search results (beam 1 and 4, cache mode).
   PARITY UNPINNED: these are restatement outputs, TF1 itself was never executed.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_torch as rt, ref_numpy as rn  # noqa: E402
from tests.common import make_hp, make_batch, perturb  # noqa: E402


def reference_scalars():
    sys.path.insert(0, "/root/reference")
    from lrs import noamlr          # reference code, imported (never copied)
    from vocab import Vocab
    out = {}
    lr = noamlr.NoamDecayLr(1.0, 0.0, 1.0, 4000, 512)
    vals = {}
    for step in (0, 1, 100, 3999, 4000, 4001, 100000):
        lr.step(step)
        vals[str(step)] = lr.get_lr()
    out["noam_lr_init1_warm4000_h512"] = vals
    lr2 = noamlr.NoamDecayLr(2.0, 1e-5, 1e-3, 400, 1024)
    vals = {}
    for step in (0, 10, 399, 400, 5000):
        lr2.step(step)
        vals[str(step)] = lr2.get_lr()
    out["noam_lr_init2_clamped_warm400_h1024"] = vals
    v = Vocab()
    out["vocab_ids"] = {"pad": v.pad(), "eos": v.eos(), "unk": v.get_id("<unk>"), "size": v.size()}
    v.insert("hello"); v.insert("world")
    out["vocab_to_id"] = v.to_id(["hello", "zzz", "world"])
    sys.path.pop(0)
    return out


def tiny_fixture(model):
    torch.set_default_dtype(torch.float64)
    rng = np.random.default_rng(11)
    hp = make_hp(model, H=16, F=32, heads=2, layers=2, Vs=13, Vt=11, max_relative_position=3, decode_length=6)
    Pn = perturb(rt.init_params(hp, model, seed=5, dtype=np.float64), rng)
    Pn = {k: v.astype(np.float32) for k, v in Pn.items()}     # stored (and consumed) as fp32
    src, tgt = make_batch(rng, 4, 7, 6, 13, 11)
    P = rt.to_torch({k: v.astype(np.float64) for k, v in Pn.items()}, torch.float64, requires_grad=True)
    out = rt.train_fn({"source": torch.tensor(src), "target": torch.tensor(tgt)}, hp, P, model, training=False)
    out["loss"].backward()
    nout = rn.loss_fn(src, tgt, hp, Pn, model)
    assert abs(float(out["loss"]) - nout["loss"]) < 1e-9
    fx = {"source": src, "target": tgt, "loss": float(out["loss"]),
          "per_sample_loss": out["per_sample_loss"].detach().numpy(),
          "gnorm": float(torch.sqrt(sum((p.grad ** 2).sum() for p in P.values())))}
    for k, v in Pn.items():
        fx["param:" + k] = v
    for k, p in P.items():
        fx["gradnorm:" + k] = float(p.grad.norm())
    Pd = rt.to_torch({k: v.astype(np.float64) for k, v in Pn.items()}, torch.float64)
    for K in (1, 4):
        hp.beam_size = K
        hp.search_mode = "cache"
        enc, dec = rt.infer_fn(hp, Pd, model)
        b = rt.beam_search({"source": torch.tensor(src)}, enc, dec, hp)
        nb = rn.beam_search(src, hp, Pn, model)
        assert np.array_equal(b["seq"], nb["seq"])
        fx["beam%d_seq" % K] = b["seq"]
        fx["beam%d_score" % K] = b["score"]
    torch.set_default_dtype(torch.float32)
    return fx


if __name__ == "__main__":
    with open(os.path.join(HERE, "reference_scalars.json"), "w") as f:
        json.dump(reference_scalars(), f, indent=1, sort_keys=True)
    for m in ("transformer", "transformer_aan", "transformer_rpr", "transformer_fuse"):
        np.savez_compressed(os.path.join(HERE, "tiny_%s.npz" % m), **tiny_fixture(m))
    print("golden fixtures written to", HERE)
