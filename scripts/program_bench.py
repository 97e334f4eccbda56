"""Encoder stack forward (6 layers, base sizes, B=64 x 64): one layer program vs launch-per-op, both replayed from a
hipGraph.  usage: python scripts/program_bench.py   (GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_params, synthetic_batch
from zero_amd.models._factory import get_core

hp = make_params(0.1)
core = get_core(hp, "transformer", None)
eng = core.eng
src, tgt = synthetic_batch(0)
batch = core.upload(src, tgt)
res = {}
for mode in (False, True, False, True):
    eng.programs_enabled = mode
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        core.encode(batch, True, True); core.encode(batch, True, True)
        torch.cuda.synchronize()
        g = eng.graph_capture(lambda: [core.encode(batch, True, True) for _ in range(5)])
        eng.graph_launch(g); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.graph_launch(g); b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 5 * 1e3
    print("programs=%s: encoder forward %.1f us (%.1f us per layer) status=%s" % (mode, us, us / 6, eng.program_status() if mode else None), flush=True)
st = eng.last_program_state.cpu().numpy()
ts = st[640:640 + 16 * 512].view(np.uint64).reshape(-1, 8).astype(np.int64)
names = ["qkv gemm", "attention", "o gemm", "add+LN", "ffn1 gemm", "ffn2 gemm", "add+LN"]
prev = ts[6, 1]
print("layer 1, workgroup 0 of group 0, shader cycles: work | vmcnt+syncthreads | atomic add | spin (polls) | tail")
for p in range(7, 14):
    t = ts[p]
    print("  %-10s work %7d | %6d | %6d | %6d (%d) | %5d   [descriptor+seed read: %d]" % (names[p - 7], t[0] - prev, t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6], t[1] - t[5], (t[7] - prev) if names[p - 7] == "add+LN" else -1))
    prev = t[1]
print("  whole program: %d cycles" % (ts[41, 0] - ts[0, 0]))
