# coding: utf-8
"""Headline benchmark: src+tgt tokens/sec of the Transformer-base training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = forward + hand-written backward + (RCCL all-reduce when N>1) + global-norm +
Adam on ONE synthetic batch per GPU: B=64 sentences x (Ls=64 + Lt=64) tokens = 8192 src+tgt
tokens per GPU-step (BASELINE.json configs[1]), V=32000, H=512, F=2048, h=8,
6+6 layers, bf16 MFMA compute / fp32 accumulate / fp32 master weights, dropouts 0.1 and label
smoothing 0.1 as in the canonical recipe.  Weak scaling: every rank draws its own batch
(seed 1234+rank).  Rank 0 prints ONE JSON line (contract in the task statement) with two
extra objects:

  roofline     -- the dominant kernel (the bf16 MFMA GEMM instance with the largest total
                  time), achieved = algorithmic FLOPs of its launches / their HIP-event time,
                  measured in an instrumented eager pass right after the timed region;
  cpu_baseline -- the oracle (oracle/ref_torch.py, unfused torch-CPU fp32 restatement of the
                  reference path: kind "port") timed on the host cores on a bounded sample of
                  the same workload.  TF1 itself cannot run here.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from zero_amd.config import transformer_base_params, SyntheticVocab  # noqa: E402
from zero_amd.utils import parallel  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md chip table (dense bf16)
B, LS, LT, V = 64, 64, 64, 32000


def synthetic_batch(rank):
    """ids ~ U{3..V-1}, last column eos(2), no padding, seed 1234+rank."""
    rng = np.random.default_rng(1234 + rank)
    src = rng.integers(3, V, size=(B, LS), dtype=np.int64)
    tgt = rng.integers(3, V, size=(B, LT), dtype=np.int64)
    src[:, -1] = 2
    tgt[:, -1] = 2
    return src, tgt


def make_params(dropout=0.1, size="base", model="transformer"):
    hp = transformer_base_params(dropout=dropout, relu_dropout=dropout, residual_dropout=dropout,
                                 attention_dropout=dropout, update_cycle=1, token_size=4096,
                                 model_name=model)
    if size == "big":        # BASELINE configs[2]: Transformer-big widths, same batch / vocabulary
        hp.override_from_dict(dict(hidden_size=1024, embed_size=1024, filter_size=4096, num_heads=16))
    hp.src_vocab = SyntheticVocab(V)
    hp.tgt_vocab = SyntheticVocab(V)
    return hp


def train_flops_per_step(hp, b=B, ls=LS, lt=LT, v=V):
    """SURVEY.md 8(d) algorithmic FLOPs: GEMM 2MNK, attention 2*2*B*Lq*Lk*H, bwd = 2x fwd."""
    H, F = hp.hidden_size, hp.filter_size
    ts, tt = b * ls, b * lt
    enc = hp.num_encoder_layer * (2 * ts * H * 3 * H + 4 * b * ls * ls * H + 2 * ts * H * H + 4 * ts * H * F)
    dec = hp.num_decoder_layer * (
        2 * tt * H * 3 * H + 4 * b * lt * lt * H + 2 * tt * H * H      # self attention
        + 2 * tt * H * H + 2 * 2 * ts * H * H + 4 * b * lt * ls * H + 2 * tt * H * H   # cross attention
        + 4 * tt * H * F)
    logits = 2 * tt * H * v
    return 3.0 * (enc + dec + logits)


class GemmProfiler(object):
    """HIP-event timing of every GEMM launch of an eager step (events on the launch stream),
    keyed by the kernel instance that runs (same names rocprofv3 --stats reports)."""

    def __init__(self, engine):
        self.eng = engine
        self.records = []
        self._gemm, self._grouped, self._kseg = engine.gemm, engine.gemm_grouped, engine.gemm_kseg

    def _name(self, M, N, K, ta, tb, out_f32, plain):
        code = self.eng.lib.raw("zk_gemm_plan")(M, N, K, out_f32, plain)
        gen, bm, bn = code & 255, (code >> 8) & 255, (code >> 16) & 255   # [27:24] split-K, [30:28] producer waves
        tf = lambda v: "true" if v else "false"
        if gen == 2:
            ns = 4 if (bm, bn) == (64, 64) else 2
            # ..., 4 compute waves, producer waves per workgroup
            return "k_gemm_dlds<%d, %d, %d, %s, %s, 4, %d>" % (bm, bn, ns, tf(ta), tf(tb), (code >> 28) & 7)
        return "k_gemm_mfma<%d, %d, %s, %s>" % (bm, bn, tf(ta), tf(tb))

    def __enter__(self):
        def timed(A, Bm, C, M, N, K, ta, tb, **kw):
            plain = not any(kw.get(k) is not None for k in ("bias", "residual")) and not kw.get("act") \
                and not kw.get("drop_p")
            name = self._name(M, N, K, ta, tb, 1 if C.t.dtype == torch.float32 else 0, 1 if plain else 0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self._gemm(A, Bm, C, M, N, K, ta, tb, **kw)
            e.record()
            self.records.append((name, 2.0 * M * N * K, s, e))

        def timed_grouped(problems, ta, tb, tile=128):
            tf = lambda v: "true" if v else "false"
            name = "k_gemm_grouped<%d, %d, %d, %s, %s, 0>" % (tile, tile, 2 if tile == 128 else 4, tf(ta), tf(tb))
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self._grouped(problems, ta, tb, tile=tile)
            e.record()
            self.records.append((name, sum(2.0 * p[3] * p[4] * p[5] for p in problems), s, e))
        def timed_kseg(segments, C, M, N, kseg, tb, residual=None):
            name = "k_gemm_kseg<64, 64, 4, %s, 4>" % ("true" if tb else "false")
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            self._kseg(segments, C, M, N, kseg, tb, residual=residual)
            e.record()
            self.records.append((name, 2.0 * M * N * kseg * len(segments), s, e))
        self.eng.gemm, self.eng.gemm_grouped, self.eng.gemm_kseg = timed, timed_grouped, timed_kseg
        return self

    def __exit__(self, *a):
        self.eng.gemm, self.eng.gemm_grouped, self.eng.gemm_kseg = self._gemm, self._grouped, self._kseg

    def calibrate(self):
        """Cost of one event pair itself: brackets around 1 and around 33 one-thread kernels,
        enqueued behind the same spin as the step.  overhead = b1 - (b33 - b1) / 32."""
        lib, st = self.eng.lib, torch.cuda.current_stream(self.eng.device)
        seed = self.eng.seed.data_ptr()
        self._cal = []
        for n in (1, 33) * 6:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(st)
            for _ in range(n):
                lib.call("zk_seed_advance", seed, 0, st.cuda_stream)
            e.record(st)
            self._cal.append((n, s, e))

    def overhead_s(self):
        torch.cuda.synchronize()
        b = {1: [], 33: []}
        for n, s, e in getattr(self, "_cal", []):
            b[n].append(s.elapsed_time(e) * 1e-3)
        if not b[1]:
            return 0.0
        med = lambda v: sorted(v)[len(v) // 2]          # medians: one late host enqueue must not move the figure
        b1, b33 = med(b[1]), med(b[33])
        return max(0.0, b1 - (b33 - b1) / 32.0)

    def summary(self):
        """{kernel: [flops, seconds, launches]}; seconds have the event-pair overhead removed."""
        ovh = self.overhead_s()
        agg = {}
        for key, fl, s, e in self.records:
            d = agg.setdefault(key, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += max(s.elapsed_time(e) * 1e-3 - ovh, 1e-7)
            d[2] += 1
        return agg


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.sh on the GPU box);
    (None, None) when no summary holds the kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(f))["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if rec:
            return rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"], os.path.basename(f)
    return None, None


def cpu_baseline(hp, budget_s=20.0):
    """Oracle (torch-CPU fp32, unfused, autograd) train step on a bounded sample."""
    from oracle import ref_torch as rt
    import copy
    hp = copy.copy(hp)
    bs = 4   # sentences of the same 64/64 shape: 512 src+tgt tokens per CPU step
    src, tgt = synthetic_batch(0)
    src, tgt = src[:bs], tgt[:bs]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(32, avail))     # more threads only add contention on these small ops
    torch.set_num_threads(cores)
    P = rt.to_torch(rt.init_params(hp, "transformer", seed=1234))
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    Vv = {k: torch.zeros_like(v) for k, v in P.items()}
    feats = {"source": torch.tensor(src), "target": torch.tensor(tgt)}
    rt.train_step(P, M, Vv, feats, hp, "transformer", 0, training=True)   # warm-up
    n, t0 = 0, time.time()
    while True:
        rt.train_step(P, M, Vv, feats, hp, "transformer", n + 1, training=True)
        n += 1
        if time.time() - t0 > budget_s or n >= 8:
            break
    dt = time.time() - t0
    return {"value": bs * (LS + LT) * n / dt, "unit": "src+tgt tokens/s", "cores": cores,
            "kind": "port",
            "sample": "%d timed steps (+1 warm-up) of %d sentences x (64+64) tokens, Transformer-base, "
                      "fwd+bwd+Adam, torch-CPU fp32 restatement of the TF1 path" % (n, bs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="transformer",
                    help="registered model name (default: the metric's model); others are side measurements")
    ap.add_argument("--size", choices=["base", "big"], default="base",
                    help="base = BASELINE configs[1] (the metric's config, default); big = configs[2]")
    args = ap.parse_args()

    rank, world, local = parallel.init_distributed()
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)

    from zero_amd.main import Trainer
    from zero_amd import hip as _hip
    for kv in os.environ.get("ZERO_HIP_TUNE", "").split(","):     # e.g. ZERO_HIP_TUNE=0:0 (A/B switches)
        if ":" in kv:
            _hip.lib().raw("zk_tune")(int(kv.split(":")[0]), int(kv.split(":")[1]))
    hp = make_params(args.dropout, args.size, args.model)
    hp.random_seed = 1234   # identical initial replicas on every rank
    tr = Trainer(hp)
    src, tgt = synthetic_batch(rank)
    tr.prepare_static({"source": src, "target": tgt})
    tr.core.eng.set_seed(1234 + rank)
    # one rank: the whole step is one hipGraph; several ranks: hipGraph segments between the
    # gradient-bucket hand-offs to RCCL (Trainer._step_segmented)
    use_graph = not args.no_graph

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):     # >= 2: eager sizing pass + graph capture
        tr.step_static(use_graph)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step_static(use_graph)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.cpu()[0])
    loss_v = float(loss.cpu()[0])
    gnorm, pnorm, skipped = tr.train_op.stats()

    # ---- roofline of the dominant kernel: instrumented eager pass (HIP events per launch).  Every
    # rank runs it (the step contains collectives); only rank 0 reports.
    # Each pass is enqueued behind a 15 ms spin kernel on the launch stream, so the launches are
    # already queued when they run and the brackets do not contain the host launch latency.
    NPROF = 3
    eng = tr.core.eng
    with GemmProfiler(eng) as prof:
        for _ in range(NPROF):
            torch.cuda.synchronize()
            eng.lib.call("zk_spin", 15000, eng.work_stream.cuda_stream)
            tr.step_static(False)
        torch.cuda.synchronize()
        with torch.cuda.stream(eng.work_stream):
            eng.lib.call("zk_spin", 15000, eng.work_stream.cuda_stream)   # long enough for the host to enqueue all of it
            prof.calibrate()
    agg = prof.summary()
    barrier()
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    tokens = world * B * (LS + LT) * args.steps
    ms = dt / args.steps * 1e3
    flops = train_flops_per_step(hp)
    out = {
        "metric": "src+tgt tokens/sec training, Transformer-%s d=%d L=6" % (args.size, hp.hidden_size),
        "value": tokens / dt, "unit": "src+tgt tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"model_name": args.model,
                   "workload": "Transformer-%s (d=%d, L=6+6, F=%d, h=%d, V=32000) training step, "
                               "B=64 x (src 64 + tgt 64) tokens per GPU, dropout %.2f, label_smooth 0.1, "
                               "fwd+bwd+allreduce+Adam" % (args.size, hp.hidden_size, hp.filter_size,
                                                           hp.num_heads, args.dropout),
                   "global_batch_tokens": world * B * (LS + LT), "parallelism": "dp%d" % world,
                   "hip_graph": ("whole step" if world == 1 else "segments between all-reduce buckets") if use_graph else False},
        "loss": loss_v, "gnorm": gnorm, "update_skipped": skipped,
        "step_mfma_frac": flops / (ms * 1e-3) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
    }
    key = max(agg, key=lambda k: agg[k][1])
    traffic, traffic_src = pmc_traffic(key)
    fl, sec, cnt = agg[key]
    tot_fl = sum(v[0] for v in agg.values())
    tot_s = sum(v[1] for v in agg.values())
    out["roofline"] = {
        "bound": "mfma", "kernel": key, "achieved": fl / sec / 1e12,
        "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS,
        "traffic": traffic, "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC)",
        "traffic_source": traffic_src, "event_pair_overhead_us": prof.overhead_s() * 1e6, "launches_per_step": cnt // NPROF, "avg_launch_us": sec / cnt * 1e6,
        "flop_per_launch": fl / cnt,
        "all_gemm_achieved": tot_fl / tot_s / 1e12, "all_gemm_ms_per_step": tot_s / NPROF * 1e3,
        "by_kernel": {k: {"launches_per_step": v[2] // NPROF, "avg_us": v[1] / v[2] * 1e6,
                          "tflops": v[0] / v[1] / 1e12} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(hp)     # bounded sample, see cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
