"""Probe for the long-K products of the feed-forward sub-layer (4096 x 512 x 2048: ffn output forward, d(ffn enlarge)):
the plain GEMM on 64x64 / 128x64 / 128x128 tiles, with and without a K split, timed as 40 launches inside one hipGraph
(events around the replay; split-K configurations include their reduction launch, reported separately from a plain
run of the same reduction)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
M, N, K = 4096, 512, 2048
REP = 40
ctx = torch.cuda.stream(e.work_stream)
ctx.__enter__()
for tb in (0, 1):
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = (torch.randn((N, K) if tb else (K, N), device="cuda") * 0.05).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for name, tile, splits, ring in (("64x64", 4, 0, 0), ("128x64", 2, 0, 0), ("128x128", 1, 0, 0), ("128x128 ring3", 1, 0, 3),
                                     ("128x128 split2", 1, 2, 0), ("128x128 split2 ring3", 1, 2, 3), ("128x64 split2", 2, 2, 0),
                                     ("64x64 split2", 4, 2, 0)):
        impl = 2 | (tile << 8) | (splits << 16) | (ring << 24)
        def body():
            for _ in range(REP):
                e.gemm(Mat(A, M, K), Mat(B, *B.shape), Mat(C, M, N), M, N, K, 0, tb, impl=impl)
        body()
        torch.cuda.synchronize()
        g = e.graph_capture(body)
        for _ in range(3):
            e.graph_launch(g)
        torch.cuda.synchronize()
        s, f = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            e.graph_launch(g)
        f.record()
        torch.cuda.synchronize()
        print("tb=%d %-22s %6.2f us per product (incl. ~1.5 us boundary%s)" %
              (tb, name, s.elapsed_time(f) * 1e3 / (5 * REP), "; + reduction launch" if splits else ""))
