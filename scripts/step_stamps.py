"""Where does a step on rotating batches lose time against a static-batch replay -- inside the step or at its boundary?
In-kernel stamps of the constant 100 MHz clock (no profiler: under rocprofv3 the two loops are equally fast) at the first
embedding launch (third node of the step's graph) and at the end of the last launch (k_norm_final2).
Build:  make -C <copy of zero_amd/csrc> STEPSTAMPS=1 ;  ZERO_HIP_LIB=<copy>/libzero_hip.so python scripts/step_stamps.py [steps]"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.config import transformer_base_params, SyntheticVocab
from zero_amd.main import Trainer
from zero_amd import hip

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hp = transformer_base_params(update_cycle=1, dropout=0.1, relu_dropout=0.1, residual_dropout=0.1, attention_dropout=0.1)
hp.src_vocab = SyntheticVocab(32000); hp.tgt_vocab = SyntheticVocab(32000)
tr = Trainer(hp)
dll = hip.lib()._dll
rng = np.random.default_rng(0)
def batch():
    s = rng.integers(3, 32000, (64, 64)); t = rng.integers(3, 32000, (64, 64)); s[:, -1] = 2; t[:, -1] = 2
    return {"source": s, "target": t}
feats = [batch() for _ in range(8)]
if os.environ.get("STAMPS_SAME") == "1":        # the rotating loop on ONE batch: is the difference the data or the loop?
    feats = [feats[0]] * 8

def read():
    buf = (ctypes.c_ulonglong * (2 * 4096))(); n2 = (ctypes.c_uint * 2)()
    assert dll.zk_step_stamps_read(buf, n2) == 0
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    return a[:n2[0]], a[4096:4096 + n2[1]]

with tr.on_work_stream():
    for i in range(40):
        tr.step(feats[i % 8])
    tr.prepare_static(feats[0])
    for i in range(3):
        tr.step_static(True)
    torch.cuda.synchronize()
    for mode in ("rotating", "static", "rotating", "static"):
        if mode == "static":
            tr.prepare_static(feats[0])
        dll.zk_step_stamps_reset()
        for i in range(steps):
            tr.step(feats[i % 8]) if mode == "rotating" else tr.step_static(True)
        torch.cuda.synchronize()
        head, tail = read()
        head = head[0::2]                       # two embedding launches per step: the source side is the first
        k = min(len(head), len(tail))
        head, tail = head[:k], tail[:k]
        inside = (tail - head)[5:] * 0.01       # us (100 MHz)
        between = (head[1:] - tail[:-1])[5:] * 0.01
        period = np.diff(head)[5:] * 0.01
        if os.environ.get("STAMPS_SEQ") == "1":
            print("   inside, us, step by step:", " ".join("%d" % v for v in (tail - head)[5:45] * 0.01))
        print("%-9s %3d steps: head->tail %.1f us (median; p10 %.1f p90 %.1f) | tail->next head %.1f us (p10 %.1f p90 %.1f) | period %.1f us"
              % (mode, k, np.median(inside), np.percentile(inside, 10), np.percentile(inside, 90), np.median(between),
                 np.percentile(between, 10), np.percentile(between, 90), np.median(period)))
