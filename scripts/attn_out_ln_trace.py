"""Phase times of the fused attention-forward launch (k_attn_out_ln) with and without the projection prologue, workgroup 0,
at the bench shape (64 sentences x 8 heads x 64 x 64, H = 512).  Build the stamped library first:
    bash scripts/build_trace.sh ATTNTRACE=1   ->  ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_trace.so python scripts/attn_out_ln_trace.py
Prints the 100 MHz-clock deltas between the ZK_AT marks of zk_attn.hip and the launch-to-launch time inside a hipGraph of 18."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
for kv in os.environ.get("ZERO_HIP_TUNE", "").split(","):     # A/B keys as in bench.py, e.g. 17:1 = row blocks fastest inside an XCD's group
    if ":" in kv:
        e.lib.raw("zk_tune")(int(kv.split(":")[0]), int(kv.split(":")[1], 0))
B, nh, L, d = 64, 8, 64, 64
H = nh * d
T = B * L
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
MARKS = {0: "start", 1: "first half stage landed (barrier 0 passed)", 2: "K loop done", 3: "projection stored, visible, barrier",
         4: "attention tile done", 5: "heads of the sentence arrived", 6: "o_map + LayerNorm tile done"}
MARKS.update({7: "  last half stage landed", 8: "  epilogue: all waves past the K loop (barrier)", 9: "  epilogue: accumulators written to the LDS",
              10: "  epilogue: barrier", 11: "  epilogue: tiles read back, packed (+ V^T pieces)", 12: "  epilogue: barrier",
              13: "  epilogue: operand tiles written"})
ORDER = [0, 1, 7, 2, 8, 9, 10, 11, 12, 13, 3, 4, 5, 6]

for pro in (3, 1, 0):
    sets = []
    for i in range(3):
        sets.append(dict(x=bf(T, H), Wp=bf(H, max(pro, 1) * H) * 0.1, bp=torch.randn(max(pro, 1) * H, device="cuda") * 0.1,
                         proj=bf(T, 3 * H), kv=bf(T, 2 * H), att=torch.empty(T, H, dtype=torch.bfloat16, device="cuda"),
                         lse=torch.zeros(B * nh * L, device="cuda"), Wo=bf(H, H) * 0.1, b=torch.randn(H, device="cuda") * 0.1,
                         R=bf(T, H), gam=torch.ones(H, device="cuda"), bet=torch.zeros(H, device="cuda"),
                         y=torch.empty(T, H, dtype=torch.bfloat16, device="cuda"), s=torch.empty(T, H, dtype=torch.bfloat16, device="cuda"),
                         mean=torch.zeros(T, device="cuda"), rstd=torch.zeros(T, device="cuda")))

    def launch(s):
        q = Mat(s["proj"], T, H, 3 * H, 0)
        if pro == 1:
            k, v = Mat(s["kv"], T, H, 2 * H, 0), Mat(s["kv"], T, H, 2 * H, H)
        else:
            k, v = Mat(s["proj"], T, H, 3 * H, H), Mat(s["proj"], T, H, 3 * H, 2 * H)
        ok = e.attn_out_ln(q, k, v, Mat(s["att"], T, H), s["lse"], B, nh, L, L, d, None, pro != 1, 0.1, 7, Mat(s["Wo"], H, H), s["b"],
                           Mat(s["R"], T, H), s["gam"], s["bet"], Mat(s["y"], T, H), Mat(s["s"], T, H), s["mean"], s["rstd"], 0.1, 8,
                           proj=(Mat(s["x"], T, H), Mat(s["Wp"], H, pro * H), s["bp"], pro) if pro else None)
        assert ok

    def body():
        e.ln_epoch_bump()
        for i in range(18):
            launch(sets[i % 3])
    with torch.cuda.stream(e.work_stream):
        body()
        torch.cuda.synchronize()
        g = e.graph_capture(body)
        for _ in range(3):
            e.graph_launch(g)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(10):
            e.graph_launch(g)
        s1.record()
        torch.cuda.synchronize()
        per = s0.elapsed_time(s1) / 10 / 18 * 1e3
        e.ln_epoch_bump()
        launch(sets[0])
        torch.cuda.synchronize()
    assert e.sync_ln_errors() == 0
    print("== projection tiles in the launch: %d: %.2f us per launch inside a hipGraph of 18" % (pro, per), flush=True)
    if hasattr(e.lib._dll, "zk_attn_trace_read"):
        buf = (ctypes.c_ulonglong * 16)()
        e.lib._dll.zk_attn_trace_read(buf)
        seen = [(i, buf[i]) for i in ORDER if buf[i]]
        t0 = buf[0]
        prev = t0
        for i, t in seen:
            if t < t0:
                continue          # mark not reached by this variant (stale value of another launch)
            print("   %-62s +%5d ns   (at %5d ns)" % (MARKS[i], (t - prev) * 10, (t - t0) * 10))
            prev = t
    if hasattr(e.lib._dll, "zk_attn_wg_trace_read"):
        import numpy as np
        n = B * nh
        wb = (ctypes.c_ulonglong * (2 * n))()
        e.lib._dll.zk_attn_wg_trace_read(wb, 2 * n)
        a = np.array(wb[:], dtype=np.int64).reshape(n, 2)
        st, en = (a[:, 0] - a[:, 0].min()) * 10, (a[:, 1] - a[:, 0].min()) * 10
        q = lambda v: "min %d p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(v, [0, 10, 50, 90, 100]).astype(int))
        print("   all %d workgroups (ns after the first one started): start %s | end %s | duration %s" % (n, q(st), q(en), q(en - st)))
        for x in range(8):
            m = (np.arange(n) & 7) == x
            print("      XCD %d: start p50 %d max %d  end p50 %d max %d" % (x, np.median(st[m]), st[m].max(), np.median(en[m]), en[m].max()))
