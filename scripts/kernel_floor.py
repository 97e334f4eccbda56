"""In-graph per-launch time of the attention and LayerNorm kernels at the bench shapes and at
scaled batch sizes (fixed cost vs work).  usage: python scripts/kernel_floor.py  (GPU box)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
NL = 24
F32 = torch.float32


def timed(body, reps=10):
    with torch.cuda.stream(e.work_stream):
        body()
        g = e.graph_capture(body)
        for _ in range(2):
            e.graph_launch(g)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            e.graph_launch(g)
        t.record()
        torch.cuda.synchronize()
    return s.elapsed_time(t) / reps / NL * 1e3


H, nh, d, L = 512, 8, 64, 64
for B in (16, 64, 256):
    T = B * L
    sets = []
    for i in range(3):
        qkv = torch.randn(T, 3 * H, device="cuda").bfloat16()
        o = torch.empty(T, H, device="cuda", dtype=torch.bfloat16)
        do = torch.randn(T, H, device="cuda").bfloat16()
        dqkv = torch.empty(T, 3 * H, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B * nh * L, device="cuda", dtype=F32)
        sets.append((Mat(qkv, T, 3 * H), Mat(o, T, H), Mat(do, T, H), Mat(dqkv, T, 3 * H), lse))
    def fwd(drop):
        def body():
            for i in range(NL):
                qkv, o, do, dqkv, lse = sets[i % 3]
                e.attn_fwd(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), o, lse,
                           B, nh, L, L, d, causal=True, drop_p=drop, sid=3)
        return body
    def bwd(drop):
        def body():
            for i in range(NL):
                qkv, o, do, dqkv, lse = sets[i % 3]
                e.attn_bwd(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), o, do, lse,
                           dqkv.cols_slice(0, H), dqkv.cols_slice(H, 2 * H), dqkv.cols_slice(2 * H, 3 * H),
                           B, nh, L, L, d, causal=True, drop_p=drop, sid=3)
        return body
    x = Mat(torch.randn(T, H, device="cuda").bfloat16(), T, H)
    y = Mat(torch.randn(T, H, device="cuda").bfloat16(), T, H)
    out = Mat(torch.empty(T, H, device="cuda", dtype=torch.bfloat16), T, H)
    sm = Mat(torch.empty(T, H, device="cuda", dtype=torch.bfloat16), T, H)
    mean = torch.empty(T, device="cuda", dtype=F32)
    rstd = torch.empty(T, device="cuda", dtype=F32)
    g = torch.ones(H, device="cuda"); b = torch.zeros(H, device="cuda")
    def ln():
        for i in range(NL):
            e.add_ln_fwd(x, y, g, b, out, sm, mean, rstd, 0.1, 5)
    print("B=%4d  attn fwd %.1f us (dropout %.1f)   attn bwd (dq+dkv) %.1f us (dropout %.1f)   add+LN fwd %.1f us" %
          (B, timed(fwd(0.0)), timed(fwd(0.1)), timed(bwd(0.0)), timed(bwd(0.1)), timed(ln)))
