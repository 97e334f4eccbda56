"""Is the rotating-batch loop host-bound?  Time to ENQUEUE n steps (no synchronisation) against the time until they are done.
usage: python scripts/host_step_time.py [steps]   (GPU box)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.config import transformer_base_params, SyntheticVocab
from zero_amd.main import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hp = transformer_base_params(update_cycle=1, dropout=0.1, relu_dropout=0.1, residual_dropout=0.1, attention_dropout=0.1)
hp.src_vocab = SyntheticVocab(32000); hp.tgt_vocab = SyntheticVocab(32000)
tr = Trainer(hp)
rng = np.random.default_rng(0)
def batch():
    s = rng.integers(3, 32000, (64, 64)); t = rng.integers(3, 32000, (64, 64)); s[:, -1] = 2; t[:, -1] = 2
    return {"source": s, "target": t}
feats = [batch() for _ in range(8)]
with tr.on_work_stream():
    for i in range(6):
        tr.step(feats[i % 8])
    torch.cuda.synchronize()
    for mode in ("rotating", "static"):
        if mode == "static":
            tr.prepare_static(feats[0])
            for i in range(3):
                tr.step_static(True)
            torch.cuda.synchronize()
        per = []
        t0 = time.perf_counter()
        for i in range(steps):
            a = time.perf_counter()
            tr.step(feats[i % 8]) if mode == "rotating" else tr.step_static(True)
            per.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        per = np.array(per) * 1e3
        print("%-9s enqueue %.3f ms/step (median call %.3f, p90 %.3f, max %.3f) | done after %.3f ms/step"
              % (mode, (t1 - t0) / steps * 1e3, np.median(per), np.percentile(per, 90), per.max(), (t2 - t0) / steps * 1e3))
