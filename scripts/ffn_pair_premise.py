"""What a sentence-scoped fusion of the feed-forward pair would have to beat: the enlarge product
(4096 x 2048 x 512) on the 128 x 128 tiles the step uses today against the same product on the 64 x 64 tiles a fused
launch is confined to (its eight workgroups per 64-row sentence block run phase 1 as four 64 x 64 tiles each), inside a
hipGraph, rotating operand sets; beside it the empty graph node and the output product + LayerNorm launch (zk_gemm_add_ln).
usage: python scripts/ffn_pair_premise.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
NL = 48


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


tiny = torch.zeros(64, device="cuda")
print("graph node floor: %.2f us" % timed(lambda: [e.lib.call("zk_zero", tiny.data_ptr(), 256, e.stream) for _ in range(NL)]))
M, H, F = 4096, 512, 2048
sets = []
for i in range(4):
    X = torch.randn(M, H, device="cuda").bfloat16()
    W1 = (torch.randn(H, F, device="cuda") * 0.05).bfloat16()
    Hh = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
    b1 = torch.randn(F, device="cuda")
    W2 = (torch.randn(F, H, device="cuda") * 0.05).bfloat16()
    Y = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    sets.append((Mat(X, M, H), Mat(W1, H, F), Mat(Hh, M, F), b1, Mat(W2, F, H), Mat(Y, M, H)))
gam, bet, b2 = torch.ones(H, device="cuda"), torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
for name, tile in (("auto (128x128)", 0), ("128x128", 1), ("128x64", 2), ("64x128", 3), ("64x64", 4)):
    def body(tile=tile):
        for i in range(NL):
            x, w1, hh, b1, w2, y = sets[i % 4]
            e.gemm(x, w1, hh, M, F, H, 0, 0, bias=b1, act=1, impl=(2 | (tile << 8) | (1 << 16)) if tile else 0)
    print("enlarge 4096x2048x512, relu, tiles %-15s %.1f us" % (name, timed(body)))


def body_ln():
    for i in range(NL):
        x, w1, hh, b1, w2, y = sets[i % 4]
        if i % 24 == 0:
            e.ln_epoch_bump()
        e.gemm_add_ln(hh, w2, M, H, F, b2, x, gam, bet, y)


print("output product + residual + LayerNorm (zk_gemm_add_ln, K = 2048): %.1f us" % timed(body_ln))
