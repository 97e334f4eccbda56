"""How far apart are the gradient slices of TWO implementations of the bf16 storage model that differ only in the order of
their fp32 sums?  (round 6; build container, CPU)

tests/test_gpu_fullsize.py holds the HIP path's gradient slices to SLICE_TOL = (8e-2, 2e-2, 2e-2, 8e-2, 8e-2) relative L2 against
the bf16-storage oracle; the measured distances are 0.1-7.0 %.  This script measures what distance the storage model has from
ITSELF: the same oracle (Cfg.store_bf16) with every matrix product accumulated in float64 and rounded once to fp32 (forward and
backward; every bf16 rounding point unchanged) against the stored fixture.  usage: python scripts/bf16_grad_noise_floor.py [base|enc12]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_torch as rt  # noqa: E402
from tests.fullsize import fullsize_hp, fullsize_batch, fullsize_params, SLICES  # noqa: E402

torch.set_num_threads(os.cpu_count())
_mm = torch.matmul


def mm64(a, b):
    return _mm(a.double(), b.double()).to(a.dtype)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


for which in sys.argv[1:] or ["base"]:
    kw = dict(num_encoder_layer=12) if which == "enc12" else {}
    gold = np.load(os.path.join(ROOT, "tests", "golden", "%s_synth_seed1234.npz" % which))
    hp = fullsize_hp(**kw)
    model = hp.model_name
    Pn = fullsize_params(hp, model)
    src, tgt = fullsize_batch()
    t0 = time.time()
    torch.matmul = mm64
    rt.Cfg.store_bf16 = True
    try:
        P = rt.to_torch(Pn, torch.float32, requires_grad=True)
        r = rt.train_fn({"source": torch.tensor(src), "target": torch.tensor(tgt)}, hp, P, model, training=False)
        r["loss"].backward()
    finally:
        torch.matmul = _mm
        rt.Cfg.store_bf16 = False
    names = [str(n) for n in gold["names"]]
    gn = np.array([float(P[k].grad.double().norm()) if P[k].grad is not None else 0.0 for k in names])
    ref = gold["bf16_grad_norms"]
    big = ref > 1e-4 * ref.max()
    print("%s (%.0f s): loss %.8f (stored bf16-storage oracle %.8f, fp32 oracle %.8f)" %
          (which, time.time() - t0, r["loss"].item(), float(gold["bf16_loss"]), float(gold["f32_loss"])))
    print("   per-variable gradient norms vs the stored bf16-storage oracle: max rel %.4f (variables above 1e-4 of the largest: %.4f)"
          % (np.max(np.abs(gn - ref) / np.maximum(ref, 1e-30)), np.max((np.abs(gn - ref) / np.maximum(ref, 1e-30))[big])))
    for i, (k, rs, cs) in enumerate(SLICES):
        mine = P[k].grad[rs[0]:rs[1], cs[0]:cs[1]].numpy()
        print("   slice %d (%s): rel L2 vs stored bf16-storage oracle %.4f, vs fp32 oracle %.4f   (stored bf16 vs fp32: %.4f)"
              % (i, k, rel(mine, gold["bf16_slice%d" % i]), rel(mine, gold["f32_slice%d" % i]),
                 rel(gold["bf16_slice%d" % i], gold["f32_slice%d" % i])), flush=True)
