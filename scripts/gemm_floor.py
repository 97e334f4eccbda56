"""Fixed cost vs per-K-step cost of the small-GEMM kernels: K sweep at M=4096, N=512 inside a
hipGraph (no host launch cost), rotating over 4 operand sets so inputs are not L1/L2-hot from the
previous launch.  Also the per-node floor of the graph (a 1-block kernel).
usage: python scripts/gemm_floor.py   (GPU box)"""
import sys, os
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
print("graph node floor (1-block memset kernel): %.2f us" % timed(lambda: [e.lib.call("zk_zero", tiny.data_ptr(), 256, e.stream) for _ in range(NL)]))
for tb in (0, 1):
    for tile in (4, 2, 1):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048):
            M, N = 4096, 512
            sets = []
            for i in range(4):
                A = torch.randn(M, K, device="cuda").bfloat16()
                B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
                C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                sets.append((Mat(A, M, K), Mat(B, *B.shape), Mat(C, M, N)))
            def body():
                for i in range(NL):
                    a, b, c = sets[i % 4]
                    e.gemm(a, b, c, M, N, K, 0, tb, impl=2 | (tile << 8) | (1 << 16))
            row.append("K=%d: %.1f" % (K, timed(body)))
        print("tb=%d tile=%s  " % (tb, {4: "64x64", 2: "128x64", 1: "128x128"}[tile]) + "  ".join(row))
