# coding: utf-8
"""Where the HOST spends a training step (Trainer.step on rotating batches): wall time inside upload / commit /
graph launch / the rest, per step, and how far the host runs ahead of the device.  bench.py's timed loop, instrumented."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from bench import make_params, synthetic_batch, ROTATION  # noqa: E402
from zero_amd.main import Trainer  # noqa: E402

hp = make_params(0.1, "base", "transformer")
hp.random_seed = 1234
tr = Trainer(hp)
feats = [dict(zip(("source", "target"), synthetic_batch(0, 64, i))) for i in range(ROTATION)]
acc = {}


def wrap(obj, name, key):
    fn = getattr(obj, name)

    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)


wrap(tr.core, "upload", "upload")
wrap(tr.core, "commit", "commit")
wrap(tr.core.eng, "graph_launch", "graph_launch")
wrap(tr.train_op, "set_hyper", "set_hyper")
for i in range(6):
    tr.step(feats[i % ROTATION])
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for mode in ("rotating", "static"):
    acc.clear()
    per = []
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    for i in range(N):
        t0 = time.perf_counter()
        if mode == "rotating":
            tr.step(feats[i % ROTATION])
        else:
            tr.step_static(True)
        per.append(time.perf_counter() - t0)
    t_host = time.perf_counter() - t_begin
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_begin
    per = np.array(per) * 1e3
    print("%s: %.3f ms/step wall; host loop done after %.1f ms of %.1f ms (%.0f %%); per-call ms: median %.3f min %.3f max %.3f"
          % (mode, t_all / N * 1e3, t_host * 1e3, t_all * 1e3, 100 * t_host / t_all, np.median(per), per.min(), per.max()))
    print("   host ms per step inside: " + ", ".join("%s %.3f" % (k, v / N * 1e3) for k, v in sorted(acc.items())))
    tr.prepare_static(feats[0])
