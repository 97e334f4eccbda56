"""Which torch calls of Trainer.step launch the small fill / copy kernels a rocprofv3 trace of the captured step shows
(profiles/r05_rocprof_captured_step.txt: ~5 FillFunctor and ~7 copyBuffer launches per step beside the graph replay)?
Runs a few rotating steps under torch's profiler and prints, for every non-library kernel / memcpy, the Python stack that
issued it.  usage (GPU box): python scripts/find_side_launches.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from zero_amd.main import Trainer  # noqa: E402

hp = bench.make_params(0.1, "base", "transformer")
hp.random_seed = 1234
tr = Trainer(hp)
feats = [dict(zip(("source", "target"), bench.synthetic_batch(0, None, i))) for i in range(bench.ROTATION)]
for i in range(4):
    tr.step(feats[i % len(feats)], True)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(4, 8):
        tr.step(feats[i % len(feats)], True)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    name = ev.name
    if name.startswith("aten::") and any(k in name for k in ("fill_", "zero_", "copy_", "zeros", "full", "_to_copy", "ones")):
        stack = [s for s in (ev.stack or []) if "zero_amd" in s or "bench.py" in s][:3]
        agg[(name, tuple(stack), str(getattr(ev, "device_type", "")))] += 1
for (name, stack, dev), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%3d x %-18s %s" % (n, name, " <- ".join(stack)))
