"""The large GEMMs of the training step (logits trio + the grouped weight gradients: 60 % of the step's FLOPs)
on the candidate tiles: 128x128 (4 waves of 64x64), 256x128 / 128x256 (4 waves of 128x64 + 4 producer waves).
Checks every variant against an fp32 torch product, then times it inside a hipGraph.
usage: python scripts/gemm_big_bench.py   (GPU box)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat

e = Engine("cuda:0")
T, V, H, F = 4096, 32000, 512, 2048
torch.manual_seed(0)


def timed(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        torch.cuda.synchronize()
        g = e.graph_capture(lambda: [fn() for _ in range(reps)])
        e.graph_launch(g); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e.graph_launch(g); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def single(name, M, N, K, ta, tb, f32):
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
    am, bm, cm = Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N)
    rows = slice(0, 512)
    ref = ((A.float().t() if ta else A.float())[rows] @ (B.float().t() if tb else B.float()))
    out = []
    for tile, ns in ((1, 0), (6, 3), (6, 2), (7, 3), (7, 2)):
        impl = 2 | (tile << 8) | (ns << 24)
        C.zero_()
        e.gemm(am, bm, cm, M, N, K, ta, tb, impl=impl)
        torch.cuda.synchronize()
        err = float((C.float()[rows] - ref).norm() / ref.norm())
        us = timed(lambda: e.gemm(am, bm, cm, M, N, K, ta, tb, impl=impl))
        out.append("%s/ns%d %7.1f us %5.0f TF err %.1e" % ({1: "128x128", 6: "256x128", 7: "128x256"}[tile], ns, us,
                                                         2.0 * M * N * K / us / 1e6, err))
    print("%-12s %s" % (name, " | ".join(out)), flush=True)


def grouped():
    X = torch.randn(T, F, device="cuda").bfloat16()
    dY = torch.randn(T, F, device="cuda").bfloat16()
    DL = torch.randn(T, V, device="cuda").bfloat16()
    probs, flops = [], 0.0
    shapes = []
    for l in range(6):      # one decoder side: qkv, o, q, k, v, o, ffn1, ffn2 per layer + the logits problem
        shapes += [(H, 3 * H), (H, H), (H, H), (H, H), (H, H), (H, H), (H, F), (F, H)]
    outs = []
    for (m, n) in shapes:
        G = torch.empty(m, n, device="cuda", dtype=torch.float32)
        outs.append(G)
        probs.append((Mat(X, T, m, F), Mat(dY, T, n, F), Mat(G, m, n), m, n, T, None))
        flops += 2.0 * m * n * T
    GE = torch.empty(V, H, device="cuda", dtype=torch.float32)
    probs.append((Mat(DL, T, V), Mat(X, T, H, F), Mat(GE, V, H), V, H, T, None))
    flops += 2.0 * V * H * T
    ref = X[:, :H].float().t() @ dY[:, :F].float()
    res = []
    for tile in (128, (128, 256), (256, 256), (256, 256, 0)):
        for g_ in outs:
            g_.zero_()
        e.gemm_grouped(probs, 1, 0, tile=tile)
        torch.cuda.synchronize()
        err = float((outs[6] - ref).norm() / ref.norm())
        refE = DL[:, :256].float().t() @ X[:, :H].float()
        errE = float((GE[:256] - refE).norm() / refE.norm())
        us = timed(lambda: e.gemm_grouped(probs, 1, 0, tile=tile), reps=4)
        res.append("%s %7.1f us %5.0f TF err %.1e %.1e" % (tile, us, flops / us / 1e6, err, errE))
    print("grouped wgrad (decoder side + logits, %.0f GFLOP): %s" % (flops / 1e9, " | ".join(res)), flush=True)


def grouped_one(name, M, N, K, ta, tb):
    """one plain fp32-output GEMM through the grouped launch, all tiles"""
    A = torch.randn((K, M) if ta else (M, K), device="cuda").bfloat16()
    B = torch.randn((N, K) if tb else (K, N), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ref = ((A.float().t() if ta else A.float())[:256] @ (B.float().t() if tb else B.float()))
    res = []
    for tile in (128, (128, 256), (256, 256), (256, 256, 0)):
        C.zero_()
        probs = [(Mat(A, *A.shape), Mat(B, *B.shape), Mat(C, M, N), M, N, K, None)]
        e.gemm_grouped(probs, ta, tb, tile=tile)
        torch.cuda.synchronize()
        err = float((C[:256] - ref).norm() / ref.norm())
        us = timed(lambda: e.gemm_grouped(probs, ta, tb, tile=tile), reps=4)
        res.append("%s %7.1f us %5.0f TF err %.1e" % (tile, us, 2.0 * M * N * K / us / 1e6, err))
    print("%-12s %s" % (name, " | ".join(res)), flush=True)


def sched_sweep():
    """Round 5: the LDS-DMA issue schedule of the 256x256 tile's second half-workgroup (tuning key 14, gemm256_acc) on the
    three launches that use the tile in the training step: all weight gradients of a step in one group (~495 GFLOP), the
    logits forward (via the grouped launch: spread form) and dlogits x E."""
    tune = e.lib.raw("zk_tune")
    X = torch.randn(T, F, device="cuda").bfloat16()
    dY = torch.randn(T, F, device="cuda").bfloat16()
    DL = torch.randn(T, V, device="cuda").bfloat16()
    probs, flops, outs = [], 0.0, []
    shapes = []
    for l in range(6):
        shapes += [(H, 3 * H), (H, H), (H, F), (F, H)]                               # encoder layer
        shapes += [(H, 3 * H), (H, H), (H, H), (H, H), (H, H), (H, H), (H, F), (F, H)]   # decoder layer
    for (m, n) in shapes:
        G = torch.empty(m, n, device="cuda", dtype=torch.float32)
        outs.append(G)
        probs.append((Mat(X, T, m, F), Mat(dY, T, n, F), Mat(G, m, n), m, n, T, None))
        flops += 2.0 * m * n * T
    GE = torch.empty(V, H, device="cuda", dtype=torch.float32)
    probs.append((Mat(DL, T, V), Mat(X, T, H, F), Mat(GE, V, H), V, H, T, None))
    flops += 2.0 * V * H * T
    ref = X[:, :H].float().t() @ dY[:, :3 * H].float()
    A = torch.randn(T, H, device="cuda").bfloat16()
    E = torch.randn(V, H, device="cuda").bfloat16()
    LG = torch.empty(T, V, device="cuda", dtype=torch.float32)
    lref = A[:256].float() @ E.float().t()
    lp = [(Mat(A, T, H), Mat(E, V, H), Mat(LG, T, V), T, V, H, None)]
    scheds = (0, 129, 130, 133, 134, 8, 16, 24)
    res = {sc: [[], [], []] for sc in set(scheds)}
    for rep in range(5):                 # round-robin, five rounds: the clocks drift by ~15 % over the first seconds of a run
        for sched in scheds:
            tune(14, sched)
            if rep == 0:
                for g_ in outs:
                    g_.zero_()
                e.gemm_grouped(probs, 1, 0, tile=(256, 256, 0))
                LG.zero_()
                e.gemm_grouped(lp, 0, 1, tile=(256, 256, 0))
                torch.cuda.synchronize()
                err = float((outs[0] - ref).norm() / ref.norm())
                lerr = float((LG[:256] - lref).norm() / lref.norm())
                assert err < 1e-5 and lerr < 1e-5, (sched, err, lerr)
            res[sched][0].append(timed(lambda: e.gemm_grouped(probs, 1, 0, tile=(256, 256, 0)), reps=4))
            res[sched][1].append(timed(lambda: e.gemm_grouped(lp, 0, 1, tile=(256, 256, 0)), reps=4))
            res[sched][2].append(timed(lambda: e.gemm_grouped(lp, 0, 1, tile=(256, 256)), reps=4))
    import statistics
    for sched in sorted(set(scheds)):
        w, l, l2 = (res[sched][i][2:] for i in range(3))          # first round dropped
        print("sched %2d: all weight gradients min %6.1f median %6.1f us (%4.0f TF) | logits fwd 256x256n min %6.1f median %6.1f us "
              "(%4.0f TF) | logits fwd spread min %6.1f median %6.1f us" %
              (sched, min(w), statistics.median(w), flops / min(w) / 1e6, min(l), statistics.median(l),
               2.0 * T * V * H / min(l) / 1e6, min(l2), statistics.median(l2)), flush=True)
    tune(14, 0)


if "--sched" in sys.argv:
    sched_sweep()
    sys.exit(0)
if "--quick" not in sys.argv:
    single("logits fwd", T, V, H, 0, 1, 1)
    single("wgrad lgt", V, H, T, 1, 0, 1)
    single("dgrad lgt", T, H, V, 0, 0, 0)
grouped_one("logits fwd", T, V, H, 0, 1)
grouped_one("wgrad lgt", V, H, T, 1, 0)
grouped_one("4096^3 NT", 4096, 4096, 4096, 0, 1)
grouped_one("4096^3 TN", 4096, 4096, 4096, 1, 0)
grouped_one("ragged TN", 1000, 520, 777 * 8, 1, 0)
grouped()
