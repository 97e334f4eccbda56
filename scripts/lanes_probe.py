"""Probe: does the latency-bound chain of a training step (small GEMMs, attention, residual + LayerNorm; ~230 launches that
do not fill the chip) run faster as P concurrent chains over B / P sentences each (P branches of ONE hipGraph, each on its
own stream) than as one chain over all B sentences?  The chain kernels are all sentence-local, so the split changes no result.
Emulates NL encoder layers forward (qkv, attention, o, add+LN, ffn in, ffn out, add+LN) at the bench shape.
usage: python scripts/lanes_probe.py   (GPU box)"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.func import Engine, Mat  # noqa: E402

e = Engine("cuda:0")
B, L, H, F, NH = 64, 64, 512, 2048, 8
NL = 6
T = B * L
dev = "cuda"
bf = torch.bfloat16


def rnd(*shape):
    return (torch.randn(*shape, device=dev) * 0.05).to(bf)


W = [dict(qkv=rnd(H, 3 * H), o=rnd(H, H), f1=rnd(H, F), f2=rnd(F, H), bq=torch.zeros(3 * H, device=dev),
          bo=torch.zeros(H, device=dev), b1=torch.zeros(F, device=dev), b2=torch.zeros(H, device=dev),
          g=torch.ones(H, device=dev), be=torch.zeros(H, device=dev)) for _ in range(NL)]
x0 = rnd(T, H)
acts = [dict(qkv=torch.empty(T, 3 * H, device=dev, dtype=bf), att=torch.empty(T, H, device=dev, dtype=bf),
             y=torch.empty(T, H, device=dev, dtype=bf), x1=torch.empty(T, H, device=dev, dtype=bf),
             h=torch.empty(T, F, device=dev, dtype=bf), x2=torch.empty(T, H, device=dev, dtype=bf),
             lse=torch.empty(B * NH * L, device=dev), s1=torch.empty(T, H, device=dev, dtype=bf),
             s2=torch.empty(T, H, device=dev, dtype=bf), m1=torch.empty(T, device=dev), r1=torch.empty(T, device=dev),
             m2=torch.empty(T, device=dev), r2=torch.empty(T, device=dev)) for _ in range(NL)]
kmask = torch.ones(B, L, device=dev)


def rows(t, b0, nb, per=L):
    return t[b0 * per:(b0 + nb) * per]


def chain(b0, nb):
    """the NL layers over sentences b0 .. b0 + nb on the current stream"""
    M = nb * L
    x = Mat(rows(x0, b0, nb), M, H)
    for l in range(NL):
        w, a = W[l], acts[l]
        qkv = Mat(rows(a["qkv"], b0, nb), M, 3 * H)
        e.gemm(x, Mat(w["qkv"], H, 3 * H), qkv, M, 3 * H, H, 0, 0, bias=w["bq"])
        att = Mat(rows(a["att"], b0, nb), M, H)
        e.attn_fwd(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att,
                   rows(a["lse"], b0, nb, NH * L), nb, NH, L, L, H // NH, kmask=rows(kmask, b0, nb, 1))
        y = Mat(rows(a["y"], b0, nb), M, H)
        e.gemm(att, Mat(w["o"], H, H), y, M, H, H, 0, 0, bias=w["bo"])
        x1 = Mat(rows(a["x1"], b0, nb), M, H)
        e.add_ln_fwd(x, y, w["g"], w["be"], x1, Mat(rows(a["s1"], b0, nb), M, H), rows(a["m1"], b0, nb), rows(a["r1"], b0, nb))
        h = Mat(rows(a["h"], b0, nb), M, F)
        e.gemm(x1, Mat(w["f1"], H, F), h, M, F, H, 0, 0, bias=w["b1"], act=1)
        e.gemm(h, Mat(w["f2"], F, H), y, M, H, F, 0, 0, bias=w["b2"])
        x2 = Mat(rows(a["x2"], b0, nb), M, H)
        e.add_ln_fwd(x1, y, w["g"], w["be"], x2, Mat(rows(a["s2"], b0, nb), M, H), rows(a["m2"], b0, nb), rows(a["r2"], b0, nb))
        x = x2


side = [torch.cuda.Stream() for _ in range(7)]


def body(P):
    if P == 1:
        chain(0, B)
        return
    main = torch.cuda.current_stream()
    nb = B // P
    fork = torch.cuda.Event()
    fork.record(main)
    for p in range(1, P):
        side[p - 1].wait_event(fork)
    # interleave the host-side issue order: op by op is not possible with a plain function per chain, and the order of
    # capture does not matter for a graph -- the branches are independent
    chain(0, nb)
    for p in range(1, P):
        with torch.cuda.stream(side[p - 1]):
            chain(p * nb, nb)
            j = torch.cuda.Event()
            j.record(side[p - 1])
        main.wait_event(j)


def timed(P, reps=20):
    with torch.cuda.stream(e.work_stream):
        body(P)
        torch.cuda.synchronize()
        g = e.graph_capture(lambda: body(P))
        nodes = e.last_graph_nodes
        for _ in range(3):
            e.graph_launch(g)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            e.graph_launch(g)
        t.record()
        torch.cuda.synchronize()
    return s.elapsed_time(t) / reps, nodes


def timed_separate(P, reps=20):
    """P graphs, one per chain, each replayed on its own stream (what the decode lanes do)"""
    nb = B // P
    streams = [e.work_stream] + side[:P - 1]
    graphs = []
    for p in range(P):
        with torch.cuda.stream(streams[p]):
            chain(p * nb, nb)
            torch.cuda.synchronize()
            graphs.append(e.graph_capture(lambda p=p: chain(p * nb, nb)))
    torch.cuda.synchronize()

    def launch_all():
        for p in range(P):
            with torch.cuda.stream(streams[p]):
                e.graph_launch(graphs[p])
    for _ in range(3):
        launch_all()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(reps):
        launch_all()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ref = None
for P in (1, 2, 4):
    print("separate graphs on %d streams: %.3f ms per replay (host clock)" % (P, timed_separate(P)), flush=True)
for P in (1, 2, 4, 8, 1, 2, 4):
    ms, nodes = timed(P)
    out = acts[NL - 1]["x2"].float().clone()
    if ref is None:
        ref = out
    same = bool((out == ref).all())
    print("chains %d: %.3f ms per replay (%d graph nodes, %.1f us per layer-chain launch), bit-identical to one chain: %s"
          % (P, ms, nodes, ms * 1e3 / (7 * NL), same), flush=True)
