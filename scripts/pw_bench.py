"""Producer-wave GEMM variants (ZERO_HIP_TUNE key 6) against the default kernels: bit-identical results and
in-graph time per GEMM for the shapes of the training step.  usage: python scripts/pw_bench.py   (GPU box)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
NL = 24


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


SHAPES = [  # M, N, K, ta, tb, tile
    (4096, 512, 512, 0, 0, 4), (4096, 512, 512, 0, 1, 4), (4096, 512, 2048, 0, 1, 4), (4096, 2048, 512, 0, 0, 2),
    (4096, 2048, 512, 0, 0, 1), (4096, 32768, 512, 0, 1, 1), (4096, 512, 32768, 0, 0, 4), (32768, 512, 4096, 1, 0, 1),
    (4000, 520, 520, 0, 0, 4), (4096, 2048, 2048, 0, 0, 1),
]
TUNES = [0, 2, 4, 8, 2 << 4, 4 << 4, 8 << 4, (4 << 4) | 256, (8 << 4) | 256, 4 << 12, 8 << 12]
for (M, N, K, ta, tb, tile) in SHAPES:
    sets = []
    for i in range(4):
        A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
        B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        sets.append((Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N)))
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").bfloat16()
    ref = None
    row = []
    for tune in TUNES:
        applies = {4: tune & 15, 1: (tune >> 4) & 15, 2: (tune >> 12) & 15}[tile] or tune == 0
        if not applies:
            continue
        e.lib.query("zk_tune", 6, tune)
        a, b, c = sets[0]
        e.gemm(a, b, c, M, N, K, ta, tb, bias=bias, residual=Mat(res, M, N), drop_p=0.1, sid=3, impl=2 | (tile << 8) | (1 << 16))
        torch.cuda.synchronize()
        out = c.t.float().clone()
        if ref is None:
            ref = out
        same = bool((out == ref).all())
        def body():
            for i in range(NL):
                a, b, c = sets[i % 4]
                e.gemm(a, b, c, M, N, K, ta, tb, impl=2 | (tile << 8) | (1 << 16))
        row.append("tune=%d: %.1f us%s" % (tune, timed(body), "" if same else " MISMATCH"))
    e.lib.query("zk_tune", 6, 0)
    print("M,N,K=%d,%d,%d ta=%d tb=%d tile=%d  " % (M, N, K, ta, tb, tile) + "  ".join(row), flush=True)
