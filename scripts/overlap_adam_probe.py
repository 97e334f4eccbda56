# coding: utf-8
"""Round-4 probe (EXPERIMENTS library): the decoder-side parameters updated on a side stream BESIDE the encoder backward
(Trainer.overlap_adam: per-side weight-gradient groups, zk_adam_range / zk_adam_finish), with the background pieces on few
blocks (tuning key 13) so that they trickle along instead of taking the chain's bandwidth and wave slots.
    ZERO_HIP_LIB=.../libzero_hip_exp.so python scripts/overlap_adam_probe.py
"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import make_params, synthetic_batch  # noqa: E402
from zero_amd import hip  # noqa: E402
from zero_amd.main import Trainer  # noqa: E402
from zero_amd.models._factory import reset_cores  # noqa: E402
from zero_amd.variables import reset_stores  # noqa: E402


def run(mode, blocks=0, steps=40):
    reset_cores(); reset_stores()
    hp = make_params(0.1, "base", "transformer")
    hp.random_seed = 1234
    tr = Trainer(hp)
    if mode != "default":
        tr.core.group_all = False
        tr.core.wgrad_tile = (128, 256)
        tr.overlap_adam = mode == "overlap"
    hip.lib().raw("zk_tune")(13, blocks)
    src, tgt = synthetic_batch(0, 64)
    tr.prepare_static({"source": src, "target": tgt})
    for _ in range(4):
        tr.step_static(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step_static(True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


assert hip.lib().experiments, "needs the EXPERIMENTS=1 library (scripts/build_experiments.sh)"
print("default (one weight-gradient launch, one Adam pass): %.3f ms" % run("default"))
print("per-side groups, Adam pass at the end:               %.3f ms" % run("split"))
for nb in (2048, 512, 256, 128, 64, 32):
    print("per-side groups, decoder-side update beside the encoder backward on %4d blocks: %.3f ms" % (nb, run("overlap", nb)))
print("default again: %.3f ms" % run("default"))
