"""Phase times of the wide residual + LayerNorm backward (k_add_ln_bwd_wide), block 0, at the bench shape (4096 x 512), with and
without residual dropout.  Needs a `make ATTNTRACE=1` library (see scripts/attn_bwd_trace.py)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
T, H = 4096, 512
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(torch.bfloat16)
MARKS = ["start", "row loaded, products formed", "two wave reductions", "row stored (loop done)", "partials in LDS, barrier",
         "partials summed and stored"]
for drop in (0.0, 0.1):
    sets = []
    for i in range(3):
        sets.append(dict(dout=bf(T, H), s=bf(T, H), mean=torch.zeros(T, device="cuda"), rstd=torch.ones(T, device="cuda"),
                         gamma=torch.ones(H, device="cuda"), dsum=bf(T, H), dy=bf(T, H),
                         ws=torch.empty(e.lib.query("zk_add_ln_bwd_workspace", T, H), dtype=torch.uint8, device="cuda")))

    def launch(x):
        m = lambda t: Mat(t, T, H)
        e.add_ln_bwd(m(x["dout"]), m(x["s"]), x["mean"], x["rstd"], x["gamma"], m(x["dsum"]), m(x["dy"]), None, None, None,
                     drop_p=drop, sid=3, private_ws=x["ws"])

    def body():
        for i in range(30):
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
        launch(sets[0])
        torch.cuda.synchronize()
    print("== dropout %.1f: %.2f us per launch inside a hipGraph of 30" % (drop, s0.elapsed_time(s1) / 10 / 30 * 1e3), flush=True)
    if hasattr(e.lib._dll, "zk_ln_trace_read"):
        buf = (ctypes.c_ulonglong * 8)()
        e.lib._dll.zk_ln_trace_read(buf)
        for i in range(1, 6):
            print("   %-40s +%5d ns   (at %5d ns)" % (MARKS[i], (buf[i] - buf[i - 1]) * 10, (buf[i] - buf[0]) * 10))
