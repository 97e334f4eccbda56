"""Phase times of the fused decode attention launch (zk_dec_cross / zk_dec_self), workgroup 0.
Build with `make -C zero_amd/csrc clean all DECTRACE=1` first (the default build has no stamps):
    python scripts/dec_attn_trace.py [self]
Prints the 100 MHz-clock deltas between the phase marks of zero_amd/csrc/zk_decfuse.hip and the launch time seen by
HIP events around a replayed hipGraph of 20 launches."""
import ctypes
import os
import sys

import numpy as np
import torch

from zero_amd import hip

lib = hip.lib()
dev = torch.device("cuda:0")
B, R, nh, H, Ls, Tmax = 32, 4, 8, 512, 30, 80      # 16-row groups: 8 x 8 workgroups
self_mode = len(sys.argv) > 1 and sys.argv[1] == "self"
g = torch.Generator(device="cpu").manual_seed(1)
bf = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
x, z, cat = bf(B * R, H), bf(B * R, 2 * H), bf(B * R, 2 * H)
xout = torch.empty_like(x)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
PW, PKV = int(os.environ.get("PAD_W", "0")), int(os.environ.get("PAD_KV", "0"))     # row padding in elements
wqt = bf(3 * H if self_mode else H, H + PW) * 0.1
wot = bf(H, H + PW) * 0.1
bq = torch.zeros(3 * H, device=dev)
kv = bf(B * Ls, 2 * H + PKV)
mask = torch.ones(B, Ls, device=dev)
kc, vc = bf(B * R, Tmax, H), bf(B * R, Tmax, H)
side = torch.cuda.Stream()
parts = torch.empty(nh, B * R, H, device=dev)
torch.cuda.set_stream(side)
st = side.cuda_stream
pro = (x.data_ptr(), None, gamma.data_ptr(), beta.data_ptr(), xout.data_ptr(), H, 1e-6, z.data_ptr(), cat.data_ptr(),
       None, 0, 0, None, None, None, 1.0, None)


def launch():
    if self_mode:
        lib.call("zk_dec_self", *pro, wqt.data_ptr(), H + PW, bq.data_ptr(), kc.data_ptr(), vc.data_ptr(), Tmax, 40, None,
                 wot.data_ptr(), H + PW, parts.data_ptr(), B, R, nh, 0.125, None, None, 0, st)
    else:
        LD = 2 * H + PKV
        lib.call("zk_dec_cross", *pro, wqt.data_ptr(), H + PW, bq.data_ptr(), kv.data_ptr(), kv.data_ptr() + H * 2, LD,
                 LD, Ls * LD, Ls * LD, mask.data_ptr(), Ls, wot.data_ptr(), H + PW, parts.data_ptr(), B, R, nh, Ls,
                 0.125, 1e9, None, None, 0, 0, None, st)


lib.raw("zk_dec_group")(int(os.environ.get("GROUP", "0")))
for _ in range(5):
    launch()
torch.cuda.synchronize()
try:
    lib.raw("zk_dec_trace_set_mode")(int(os.environ.get("MODE", "0")))
    rd = lib.raw("zk_dec_trace_read")
    rd.argtypes = [ctypes.c_void_p]
    buf = (ctypes.c_ulonglong * 16)()
    rows = []
    for _ in range(20):
        launch()
        torch.cuda.synchronize()
        rd(buf)
        rows.append([buf[i] for i in (0, 1, 8, 9, 10, 11, 2, 3, 4, 5, 6, 7)])
    d = np.diff(np.array(rows, dtype=np.float64), axis=1) * 10.0        # ns
    names = ["setup", "prologue loads issued", "prefetch issued", "LayerNorm computed", "rows -> LDS", "barrier",
             "projection MFMA + combine", "scores", "softmax", "context", "output projection + stores"]
    for n, v in zip(names, np.median(d, axis=0)):
        print("%-28s %7.0f ns" % (n, v))
    print("%-28s %7.0f ns (in-kernel span of workgroup 0)" % ("total", np.median(d.sum(axis=1))))
except AttributeError:
    print("library built without DECTRACE=1: no phase stamps")
gexec = None
lib.call("zk_graph_begin", st)
for _ in range(20):
    launch()
out = ctypes.c_void_p()
lib.call("zk_graph_end", st, ctypes.byref(out))
for _ in range(3):
    lib.call("zk_graph_launch", out, st)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    lib.call("zk_graph_launch", out, st)
e.record()
torch.cuda.synchronize()
print("GROUP=%s MODE=%s" % (os.environ.get("GROUP", "0"), os.environ.get("MODE", "0")), "PAD_W=%d PAD_KV=%d graph of 20 launches: %.2f us per launch" % (PW, PKV, s.elapsed_time(e) * 1e3 / 200))
