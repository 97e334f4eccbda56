"""Where do the ~60 us of the target-side zk_embed_bwd_sorted go?  (bench shapes: 4096 token rows, ~3850 distinct ids of
32000, H = 512.)  Times the call with / without accumulation into the table gradient, after the table was written by
streaming stores (as the softmax weight-gradient GEMM leaves it) or by ordinary ones."""
import numpy as np
import torch

from zero_amd import hip
from zero_amd.func import Engine

e = Engine(torch.device("cuda:0"))
T, V, H = 4096, 32000, 512
rng = np.random.default_rng(0)
ids = rng.integers(3, V, T)
order = np.argsort(ids, kind="stable")
uid, start = np.unique(ids[order], return_index=True)
seg = np.append(start, T).astype(np.int32)
sort = {"rows": torch.from_numpy(order.astype(np.int32)).cuda(), "seg": torch.from_numpy(seg).cuda(),
        "uid": torch.from_numpy(uid.astype(np.int32)).cuda(), "n": torch.tensor([len(uid)], dtype=torch.int32).cuda(),
        "max_uniq": T}
dout = e.mat("d", T, H)
dout.t.normal_()
table = torch.zeros(V, H, device="cuda")
other = torch.zeros(64 << 20, device="cuda")          # 256 MB: flushes the caches between runs


def run(acc, drop, flush):
    ts = []
    for _ in range(12):
        if flush != 2:
            table.zero_()
        if flush:
            other.add_(1.0)                # evicts the table (and the sort arrays, dout) from L2 / MALL
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        e.embed_bwd_sorted(sort, dout, table, H, accumulate=acc, drop_p=drop)
        t.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(t) * 1e3)
    return float(np.median(ts))


for acc in (False, True):
    for drop in (0.0, 0.1):
        for flush in (0, 1, 2):
            print("accumulate=%d dropout=%.1f cold=%d : %6.1f us" % (acc, drop, flush, run(acc, drop, flush)))
