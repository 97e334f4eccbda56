"""The fp32 decode mode's products (zk_f32_gemm, 128 rows) inside a hipGraph: microseconds per launch by shape, kernel
generation (zk_f32_gemm_legacy 1 / 2 / 0) and leading dimension of the activation operand (power-of-two row strides put the
sixteen rows of an A tile on one L2 channel).  Weights rotate over 6 sets (a decoder's six layers: 4 MB each, not L2-hot).
usage: python scripts/f32_gemm_bench.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine

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
print("graph node floor (1-block kernel): %.2f us" % timed(lambda: [e.lib.call("zk_zero", tiny.data_ptr(), 256, e.stream) for _ in range(NL)]))
SHAPES = [("q/o 512x512", 128, 512, 512, 0), ("ffn1 512->2048", 128, 2048, 512, 0), ("z 1024x1024", 128, 1024, 1024, 0),
          ("ffn2 2048->512", 128, 512, 2048, 0), ("logits 32000", 128, 32000, 512, 1)]
pads = [0, 32] if len(sys.argv) < 2 else [int(x) for x in sys.argv[1].split(",")]
NSETS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for name, M, N, K, tb in SHAPES:
    for mode in (1, 2, 0):
        row = []
        for pad in pads:
            lda = K + pad
            sets = []
            for i in range(NSETS):
                A = torch.randn(M, lda, device="cuda")
                B = torch.randn((N, K) if tb else (K, N), device="cuda")
                C = torch.empty(M, N, device="cuda")
                bias = torch.randn(N, device="cuda")
                sets.append((A, B, C, bias))

            def body():
                for i in range(NL):
                    A, B, C, bias = sets[i % NSETS]
                    e.lib.call("zk_f32_gemm", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, K if tb else N, N, tb,
                               bias.data_ptr(), 0, e.stream)
            e.lib.raw("zk_f32_gemm_legacy")(mode)
            try:
                row.append("lda=K+%d: %.1f us" % (pad, timed(body)))
            finally:
                e.lib.raw("zk_f32_gemm_legacy")(0)
        print("%-16s %-22s %s" % (name, {1: "round 5", 2: "32x32 K-sliced", 0: "default (16x16)"}[mode], "   ".join(row)))
