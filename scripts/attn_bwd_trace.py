"""Phase times of the single-tile attention backward kernels, workgroup (0, 0, 0), at the bench shape (64 sentences x 8 heads
x 64 x 64).  Build the stamped library first (the default build has no stamps):
    make -C zero_amd/csrc clean all ATTNTRACE=1        (or into a copy, then ZERO_HIP_LIB=<copy>/libzero_hip.so)
    python scripts/attn_bwd_trace.py
Variants: plain (dO read), oproj (dO = dY . W_o^T computed in the prologue), rpr (72-KB relative-position kernel), rpr-resident.
Prints the 100 MHz-clock deltas between the marks of zero_amd/csrc/zk_attn_dev.h (ZK_AT) and the launch-to-launch time of a
replayed hipGraph of 18 launches."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
for kv in os.environ.get("ZERO_HIP_TUNE", "").split(","):     # A/B keys as in bench.py, e.g. 15:4 = head-major grid
    if ":" in kv:
        e.lib.raw("zk_tune")(int(kv.split(":")[0]), int(kv.split(":")[1], 0))
B, nh, L, d = 64, 8, 64, 64
H = nh * d
T = B * L
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
MARKS = {0: "start", 1: "prologue loads issued (oproj: chunk loop starts)", 2: "oproj product done", 3: "operand tiles stored",
         4: "first barrier passed", 5: "phase 1 (P, dS) computed", 6: "barrier", 7: "dS / P^T (+ transposed operands) stored, barrier",
         8: "dQ / dK / dV products", 11: "bucket sums", 12: "table products", 9: "outputs staged, barrier", 10: "stores issued"}
ORDER = [0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 9, 10]

for variant in (sys.argv[1:] or ("plain", "oproj", "rpr", "rpr-resident")):
    rpr = variant.startswith("rpr")
    e.rpr_bwd_resident = variant == "rpr-resident"
    sets = []
    for i in range(3):
        q, k, v, out = bf(T, H), bf(T, H), bf(T, H), torch.zeros(T, H, dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros(B * nh * L, device="cuda")
        rk = bf(33, d) if rpr else None
        rv = bf(33, d) if rpr else None
        e.attn_fwd(Mat(q, T, H), Mat(k, T, H), Mat(v, T, H), Mat(out, T, H), lse, B, nh, L, L, d, rpr_k=rk, rpr_v=rv, max_rel=16)
        sets.append(dict(q=q, k=k, v=v, out=out, lse=lse, rk=rk, rv=rv, dy=bf(T, H), Wo=bf(H, H) * 0.1, dout=bf(T, H),
                         dq=torch.empty_like(q), dk=torch.empty_like(q), dv=torch.empty_like(q),
                         drk=torch.zeros(33, d, device="cuda") if rpr else None, drv=torch.zeros(33, d, device="cuda") if rpr else None))

    def launch(s):
        m = lambda t: Mat(t, T, H)
        e.attn_bwd(m(s["q"]), m(s["k"]), m(s["v"]), m(s["out"]), m(s["dout"]), s["lse"], m(s["dq"]), m(s["dk"]), m(s["dv"]),
                   B, nh, L, L, d, rpr_k=s["rk"], rpr_v=s["rv"], drpr_k=s["drk"], drpr_v=s["drv"], max_rel=16,
                   oproj=(m(s["dy"]), Mat(s["Wo"], H, H)) if variant == "oproj" else None)

    def body():
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
        launch(sets[0])
        torch.cuda.synchronize()
    print("== %s: %.2f us per launch inside a hipGraph of 18" % (variant, per), flush=True)
    if hasattr(e.lib._dll, "zk_attn_trace_read"):
        buf = (ctypes.c_ulonglong * 16)()
        e.lib._dll.zk_attn_trace_read(buf)
        seen = [(i, buf[i]) for i in ORDER if buf[i]]
        t0 = seen[0][1]
        prev = t0
        for i, t in seen:
            if t < t0:
                continue          # mark not reached by this variant (stale value of another launch)
            print("   %-62s +%5d ns   (at %5d ns)" % (MARKS[i], (t - prev) * 10, (t - t0) * 10))
            prev = t
