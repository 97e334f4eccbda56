"""Soak test of the real-data training entry point: many batch shapes in random order through Trainer.step
(shape-keyed hipGraph cache, buffer growth, LRU eviction): memory must plateau, steps must stay fast.
usage: python scripts/soak_train.py [steps]   (GPU box)"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.config import transformer_base_params, SyntheticVocab
from zero_amd.main import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
use_graph = os.environ.get("SOAK_GRAPH", "1") != "0"
check_every = int(os.environ.get("SOAK_CHECK", "100"))
dp = float(os.environ.get("SOAK_DROPOUT", "0.1"))
hp = transformer_base_params(update_cycle=1, dropout=dp, relu_dropout=dp, residual_dropout=dp, attention_dropout=dp)
hp.src_vocab = SyntheticVocab(32000); hp.tgt_vocab = SyntheticVocab(32000)
tr = Trainer(hp)
tr.MAX_GRAPHS = int(os.environ.get("SOAK_MAX_GRAPHS", "24"))     # small: force evictions
rng = np.random.default_rng(0)
shapes = [(int(4096 // L), L + int(rng.integers(-3, 4)), L + int(rng.integers(-3, 4))) for L in range(12, 100, 2)]
def batch(B, Ls, Lt):
    s = rng.integers(3, 32000, (B, Ls)); t = rng.integers(3, 32000, (B, Lt)); s[:, -1] = 2; t[:, -1] = 2
    return {"source": s, "target": t}
t0 = time.perf_counter(); mem = []; seen = {}
for i in range(steps):
    sh = shapes[int(rng.integers(0, len(shapes)))]
    key = sh
    before = tr._graphs.get(key)
    mode = "eager" if before is None else ("capture" if before == "warm" else "replay")
    hist = seen.setdefault(sh, [])
    hist.append((i + 1, mode))
    loss = tr.step(batch(*sh), use_graph=use_graph)
    if os.environ.get("SOAK_TRACE"):
        g_, p_, b_ = tr.train_op.stats()
        print("TRACE %d %s %s %.9g %.9g" % (i + 1, sh, mode if use_graph else "eager", float(loss.reshape(-1)[0].cpu()), g_))
    if check_every < 100:
        g, p, bad = tr.train_op.stats()
        if bad or not np.isfinite(g):
            print("first non-finite gradient norm at step", i + 1, "shape", sh, "gnorm", g, "loss", float(loss.reshape(-1)[0].cpu()))
            print("history of this shape:", hist[-6:], "graphs cached:", len(tr._graphs))
            G = tr.store.export("grad")
            for name, gval in G.items():
                bad_n = int((~np.isfinite(gval)).sum())
                if bad_n or np.abs(gval[np.isfinite(gval)]).max(initial=0) > 1e6:
                    print("   grad", name, gval.shape, "non-finite:", bad_n, "max finite:", float(np.abs(gval[np.isfinite(gval)]).max(initial=0)))
            break
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        g, p, bad = tr.train_op.stats()
        mem.append(torch.cuda.memory_allocated() / 2**20)
        print("step %4d  %.2f ms/step  loss %.3f  gnorm %.3f  alloc %.0f MiB  graphs %d  realloc_gen %d" %
              (i + 1, (time.perf_counter() - t0) / 100 * 1e3, float(loss.reshape(-1)[0].cpu()), g, mem[-1],
               len(tr._graphs), tr.core.eng.realloc_gen))
        assert not bad
        t0 = time.perf_counter()
assert mem[-1] <= mem[len(mem) // 2] * 1.02 + 1, "device memory keeps growing: %s" % mem
print("ok")
