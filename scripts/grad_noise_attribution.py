"""Which bf16 storage point puts the direction noise into the deep gradient slices?  

CPU only, the oracle at the BASELINE size (tests/fullsize.py: Transformer-base, B = 64 x (64 + 64), V = 32000): the fp32
run, the run under the full bf16 storage model (oracle/ref_torch.py Cfg.store_bf16) and one run per storage SITE with
only that site rounded (Cfg.store_sites) -- weights, outputs of linear layers, softmax probabilities, gradients of the
scores, attention outputs, LayerNorm outputs, embeddings, logits gradient.  For the five gradient slices of the
fixtures (tests/fullsize.SLICES) it prints the relative L2 distance to the fp32 gradient, per site.

    python scripts/grad_noise_attribution.py            # ~10 oracle steps of ~1 min each on the build container
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_torch as rt  # noqa: E402
from tests.fullsize import fullsize_hp, fullsize_batch, fullsize_params, SLICES  # noqa: E402

SITES = ["weights", "linear", "probs", "scores", "attn_out", "ln", "embed", "logits"]


def run(Pn, hp, src, tgt, sites):
    rt.Cfg.store_bf16 = sites != "fp32"
    rt.Cfg.store_sites = None if sites in ("fp32", "all") else set(sites)
    try:
        P = rt.to_torch(Pn, torch.float32, requires_grad=True)
        r = rt.train_fn({"source": torch.tensor(src), "target": torch.tensor(tgt)}, hp, P, hp.model_name, training=False)
        r["loss"].backward()
    finally:
        rt.Cfg.store_bf16, rt.Cfg.store_sites = False, None
    out = {"loss": float(r["loss"])}
    for i, (k, rs, cs) in enumerate(SLICES):
        out["slice%d" % i] = P[k].grad[rs[0]:rs[1], cs[0]:cs[1]].double().numpy().copy()
    out["gnorm"] = float(np.sqrt(sum(float(v.grad.double().pow(2).sum()) for v in P.values() if v.grad is not None)))
    return out


def main():
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    hp = fullsize_hp()
    Pn = fullsize_params(hp, hp.model_name)
    src, tgt = fullsize_batch()
    res = {}
    for name in ["fp32", "all"] + SITES:
        t0 = time.time()
        res[name] = run(Pn, hp, src, tgt, name if name in ("fp32", "all") else [name])
        print("%-9s loss %.6f gnorm %.5f (%.0f s)" % (name, res[name]["loss"], res[name]["gnorm"], time.time() - t0), flush=True)
    ref = res["fp32"]
    table = {}
    print("\nrelative L2 distance of the gradient slices to the fp32 oracle, one storage site rounded at a time")
    print("%-9s " % "site" + " ".join("%22s" % ("slice%d %s" % (i, SLICES[i][0].split("/")[0] + "/" + SLICES[i][0].split("/")[1] if "/" in SLICES[i][0] else SLICES[i][0])[:22]) for i in range(len(SLICES))))
    for name in ["all"] + SITES:
        row = []
        for i in range(len(SLICES)):
            a, b = res[name]["slice%d" % i], ref["slice%d" % i]
            row.append(float(np.linalg.norm(a - b) / np.linalg.norm(b)))
        table[name] = row
        print("%-9s " % name + " ".join("%22.4f" % v for v in row))
    json.dump({"slices": [s[0] for s in SLICES], "rel_l2_vs_fp32": table,
               "loss": {k: v["loss"] for k, v in res.items()}, "gnorm": {k: v["gnorm"] for k, v in res.items()}},
              open(os.path.join(ROOT, "profiles", "r03_grad_noise_attribution.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
