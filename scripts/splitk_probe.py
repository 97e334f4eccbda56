"""VERDICT r02 item 1c probe: the K = 2048 -> N = 512 products of the step (FFN output forward, FFN enlarge dgrad; 24 launches,
~21 us each on 64x64 tiles) on bigger tiles with split-K, measured inside a hipGraph with rotating operand sets
(scripts/gemm_floor.py method).  Split variants INCLUDE the slab reduction launch (k_splitk_reduce) -- an upper bound
of what a consumer-side sum of the partial products (in the LayerNorm launch) would cost on the GEMM side.
usage: python scripts/splitk_probe.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

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


TILE = {4: "64x64", 2: "128x64", 1: "128x128", 3: "64x128"}
for (M, N, K) in ((4096, 512, 2048), (4096, 512, 512), (4096, 512, 1536)):
    for tb in (0, 1):
        row = []
        for tile, split in ((4, 1), (2, 1), (2, 2), (1, 1), (1, 2), (1, 4), (3, 2)):
            sets = []
            for i in range(4):
                A = torch.randn(M, K, device="cuda").bfloat16()
                B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
                C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
                sets.append((Mat(A, M, K), Mat(B, *B.shape), Mat(C, M, N)))

            def body():
                for i in range(NL):
                    a, b, c = sets[i % 4]
                    e.gemm(a, b, c, M, N, K, 0, tb, impl=2 | (tile << 8) | (split << 16))
            try:
                row.append("%s/s%d %.1f" % (TILE[tile], split, timed(body)))
            except Exception as exc:      # noqa: BLE001
                row.append("%s/s%d ERR" % (TILE[tile], split))
        print("M,N,K=%d,%d,%d tb=%d  " % (M, N, K, tb) + "  ".join(row), flush=True)
