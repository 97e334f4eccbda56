"""Micro-benchmark of zk_gemm on the GEMM shapes of one Transformer-base training step.
usage: python scripts/gemm_bench.py [variant ...]   (GPU box)"""
import sys, os, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
T, V = 4096, 32000
H = int(os.environ.get("GEMM_BENCH_H", "512"))      # 1024 = Transformer-big widths
F = 4 * H
SHAPES = [  # name, M, N, K, ta, tb, out_f32, count per step
    ("fwd qkv   ", T, 3 * H, H, 0, 0, 0, 12), ("fwd HxH   ", T, H, H, 0, 0, 0, 36), ("fwd ffn1  ", T, F, H, 0, 0, 0, 12),
    ("fwd ffn2  ", T, H, F, 0, 0, 0, 12), ("logits    ", T, V, H, 0, 1, 1, 1),
    ("dgrad HxH ", T, H, H, 0, 1, 0, 36), ("dgrad qkv ", T, H, 3 * H, 0, 1, 0, 12), ("dgrad ffn2", T, F, H, 0, 1, 0, 12),
    ("dgrad ffn1", T, H, F, 0, 1, 0, 12), ("dgrad lgt ", T, H, V, 0, 0, 0, 1),
    ("wgrad HxH ", H, H, T, 1, 0, 1, 36), ("wgrad qkv ", H, 3 * H, T, 1, 0, 1, 12), ("wgrad ffn1", H, F, T, 1, 0, 1, 12),
    ("wgrad ffn2", F, H, T, 1, 0, 1, 12), ("wgrad lgt ", V, H, T, 1, 0, 1, 1),
]
TILES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x128", 4: "64x64", 5: "256x128"}


def run(M, N, K, ta, tb, f32, impl, reps=20):
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    am, bm, cm = Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N)
    for _ in range(3):
        e.gemm(am, bm, cm, M, N, K, ta, tb, impl=impl)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        e.gemm(am, bm, cm, M, N, K, ta, tb, impl=impl)
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps * 1e3   # us


variants = [(0, 0, 0), (0, 64, 0)] + [(t, 0, 0) for t in (1, 2, 3, 4, 5)]   # f=64: generation 1 kernel
extra_split = [(0, 0, s) for s in (1, 2, 4, 8)]
print("%-11s %-22s | " % ("shape", "M,N,K") + " | ".join("%12s" % ("%s%s" % (TILES[t], "/gen1" if f & 64 else "")) for t, f, _ in variants))
tot = {i: 0.0 for i in range(len(variants))}
for name, M, N, K, ta, tb, f32, cnt in SHAPES:
    row = []
    for i, (t, f, sp) in enumerate(variants):
        us = run(M, N, K, ta, tb, f32, (3 if f & 64 else 2) | (t << 8) | ((f & 3) << 12) | (sp << 16))
        tot[i] += us * cnt
        row.append("%6.1f %5.0f" % (us, 2.0 * M * N * K / us / 1e6))
    line = "%-11s %-22s | " % (name, "%d,%d,%d" % (M, N, K)) + " | ".join(row)
    if ta == 1 and N * M <= 2048 * 2048:
        line += " || split " + " ".join("%d:%.1f" % (sp, run(M, N, K, ta, tb, f32, 2 | (sp << 16))) for _, _, sp in extra_split)
    print(line)
print("per-step GEMM total (ms): " + " | ".join("%.2f" % (tot[i] / 1e3) for i in range(len(variants))))
