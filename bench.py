# coding: utf-8
"""Headline benchmark: src+tgt tokens/sec of the Transformer-base training step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = what a user's training loop runs per batch (zero_amd.main.Trainer.step, the call of zero_amd.main.train):
the batch's ids go host -> HBM through pinned staging slots, ONE launch derives what depends on the ids alone (masks,
loss weights, the token rows grouped by embedding id), then forward + hand-written backward + (RCCL all-reduce when
N>1) + global-norm + Adam.  The timed loop feeds a ROTATION of 8 distinct synthetic batches per GPU (seeds
1234 + rank + 1000 i), each B=64 sentences x (Ls=64 + Lt=64) tokens = 8192 src+tgt tokens per GPU-step
(BASELINE.json configs[1]), V=32000, H=512, F=2048, h=8, 6+6 layers, bf16 MFMA compute / fp32
accumulate / fp32 master weights, dropouts 0.1 and label smoothing 0.1 as in the canonical recipe.  Weak scaling: every
rank draws its own batches.  `static_batch_ms_per_step` in the same line is the old measurement (one pre-uploaded batch
replayed: the device work alone); --static-batch makes it the headline of a SIDE measurement.  With --gpus N and no launcher (WORLD_SIZE unset) bench.py starts its N ranks itself
(torch.distributed.run on 127.0.0.1); under torchrun it goes straight on.  Rank 0 prints ONE JSON line
(contract in the task statement) with these extra objects:

  roofline     -- the dominant kernel CLASS (the class with the largest time per step: splitting or merging template
                  instances cannot move it), achieved = algorithmic FLOPs of its launches / their HIP-event time,
                  measured in an instrumented eager pass right after the timed region; `by_class` holds every class
                  (small-GEMM chain, big GEMMs, attention, LayerNorm, cross entropy, Adam, other) with launches,
                  us per step, achieved TF or TB/s and the fraction of its bound; `worst_instance` / `by_kernel` the
                  kernel instances; `traffic` (HBM bytes per launch) and `mfma_busy` (MFMA-pipe utilisation) come from
                  the newest committed counter summaries (profiles/*_pmc_traffic.json, *_pmc_mfma.json: separate
                  rocprofv3 --pmc passes, stamped with their commit);
  rccl         -- ranks, distinct devices seen, and with N > 1 one short timed LEG per exchange mode in the same process
                  group ({fp32, bf16} buckets x {torch.distributed, zk_comm} x {row-sparse, dense} source-embedding
                  gradient): ms per step, bytes per rank, exposed ms; the headline is the fastest reference-exact
                  (fp32) leg unless a bf16 leg wins by more than 3 %;
  decode       -- BASELINE configs[3] (transformer_aan, beam 4, 3000 synthetic sentences, eval batch 32) under
                  the same clock: sentences/s, ms per decode step, launches per step, HBM roofline fraction,
                  batches in flight, its own cpu_baseline (one rank only);
  cpu_baseline -- the oracle (oracle/ref_torch.py, unfused torch-CPU fp32 restatement of the
                  reference path: kind "port") timed on the host cores on a bounded sample of
                  the same workload (3 warm-up + 10 timed steps when they fit 90 s).  TF1 itself cannot
                  run here.

Side measurements (never the headline): --sentences-per-gpu 256, --size big, --model ..., --mode decode.
"""

import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the HIP runtime initialises: the decode leg runs 4 batches in flight (zero_amd/evalu.py decode_many)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from zero_amd.config import transformer_base_params, SyntheticVocab  # noqa: E402
from zero_amd.utils import parallel  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md chip table (dense bf16)
B, LS, LT, V = 64, 64, 64, 32000


ROTATION = 8     # distinct batches the timed loop cycles through


def synthetic_batch(rank, b=None, i=0):
    """ids ~ U{3..V-1}, last column eos(2), no padding, seed 1234 + rank + 1000 i (batch i of the
    rank's rotation; i = 0 is the batch rounds 1-3 replayed)."""
    b = B if b is None else b
    rng = np.random.default_rng(1234 + rank + 1000 * i)
    src = rng.integers(3, V, size=(b, LS), dtype=np.int64)
    tgt = rng.integers(3, V, size=(b, LT), dtype=np.int64)
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


HBM_PEAK_TBS = 8.0               # same guide: HBM3E peak

# kernel classes of the step (roofline.by_class): name -> (bound, what it holds)
CLASSES = (
    ("small_gemm_chain", "mfma", "linear forward + dgrad on the 4096-row activations (k_gemm_dlds tiles; 58 of them carry a "
                                 "residual + LayerNorm forward or backward in their epilogue, 18 of those also the "
                                 "attention forward of their sub-layer AND the qkv_map / q_map projection in front of it "
                                 "(round 6)), grouped K/V projections, K-segmented d(encoder output)"),
    ("big_gemms", "mfma", "all weight gradients (one grouped launch of 256x256 tiles), logits forward, dlogits x E"),
    ("attention", "hbm", "attention backward, one (sentence, head) tile per workgroup, incl. the folded o_map dgrad (the "
                         "forward runs inside the output-projection launches of small_gemm_chain)"),
    ("layernorm", "hbm", "residual + LayerNorm launches that remain (the backward at the top of each stack; the other 58 of "
                         "60 run inside small_gemm_chain launches)"),
    ("cross_entropy", "hbm", "label-smoothed cross entropy: fp32 logits in, bf16 dlogits out"),
    ("adam", "hbm", "TF1 Adam + bf16 shadow refresh + norms, 30 B / parameter"),
)


class LaunchProfiler(object):
    """HIP-event timing of the launches of an eager step (events on the launch stream).  GEMM launches are keyed by the
    kernel instance that runs (same names rocprofv3 --stats reports); every wrapped launch is also booked to its
    kernel CLASS with its algorithmic FLOPs and HBM bytes (roofline.by_class)."""

    def __init__(self, engine, train_op=None):
        self.eng, self.top = engine, train_op
        self.records = []          # (class, kernel-or-None, flops, bytes, start, end)
        self._saved = {}

    def _name(self, M, N, K, ta, tb, out_f32, plain):
        code = self.eng.lib.raw("zk_gemm_plan")(M, N, K, out_f32, plain)
        gen, bm, bn = code & 255, ((code >> 8) & 255) * 8, ((code >> 16) & 255) * 8   # [27:24] split-K, [30:28] producer waves
        tf = lambda v: "true" if v else "false"
        if gen == 2:
            ns = 4 if (bm, bn) == (64, 64) else 3 if max(bm, bn) == 256 else 2
            # ..., 4 compute waves, producer waves per workgroup
            # ..., lazy-LayerNorm epilogue form (0: none; the EXPERIMENTS-only zk_gemm_ln uses 1 / 2)
            return "k_gemm_dlds<%d, %d, %d, %s, %s, 4, %d, 0>" % (bm, bn, ns, tf(ta), tf(tb), (code >> 28) & 7)
        return "k_gemm_mfma<%d, %d, %s, %s>" % (bm, bn, tf(ta), tf(tb))

    def _timed(self, cls, name, flops, nbytes, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((cls, name, float(flops), float(nbytes), s, e))
        return out

    def __enter__(self):
        eng = self.eng
        keep = self._saved
        for k in ("gemm", "gemm_grouped", "gemm_grouped_update", "gemm_kseg", "attn_fwd", "attn_bwd", "add_ln_fwd",
                  "add_ln_bwd", "ce_fused", "gemm_add_ln", "gemm_ln_bwd", "attn_out_ln",
                  "gemm_ln", "ln_fold", "add_ln_bwd_lazy"):      # (the last three: the EXPERIMENTS-only LayerNorm-free forward)
            keep[k] = getattr(eng, k)
        esz = lambda m: m.t.element_size()

        def gemm(A, Bm, C, M, N, K, ta, tb, **kw):
            plain = not any(kw.get(k) is not None for k in ("bias", "residual")) and not kw.get("act") \
                and not kw.get("drop_p")
            name = self._name(M, N, K, ta, tb, 1 if C.t.dtype == torch.float32 else 0, 1 if plain else 0)
            cls = "big_gemms" if max(N, K) > 8192 else "small_gemm_chain"
            nbytes = (M * K + K * N) * 2 + M * N * esz(C) + (M * N * 2 if kw.get("residual") is not None else 0) \
                + (M * N * 2 if kw.get("aux") is not None else 0)
            return self._timed(cls, name, 2.0 * M * N * K, nbytes, lambda: keep["gemm"](A, Bm, C, M, N, K, ta, tb, **kw))

        def gemm_grouped(problems, ta, tb, tile=128):
            tf = lambda v: "true" if v else "false"
            bm_, bn_ = (tile, tile) if isinstance(tile, int) else tile[:2]
            if (bm_, bn_) == (256, 256):
                # <TA, TB, SPREAD, CS, K32> as zk_gemm_grouped instantiates it: CS = bias column sums ride along (a problem
                # carries a colsum pointer), K32 = the 32-deep ring of the EXPERIMENTS build
                k32 = len(tile) == 3 and tile[2] == "k32"
                cs = bool(ta) and not tb and any(len(p) > 8 and p[8] is not None for p in problems)
                spread = not k32 and not (len(tile) == 3 and not tile[2])
                # ..., UPD = the optimiser update inside the launch (gemm_grouped_update below)
                name = "k_gemm_grouped256<%s, %s, %s, %s, %s, false>" % (tf(ta), tf(tb), tf(spread), tf(cs), tf(k32))
            else:
                name = "k_gemm_grouped<%d, %d, %d, %s, %s, %d>" % (bm_, bn_, 4 if bm_ == 64 else 3 if max(bm_, bn_) == 256 else 2,
                                                                  tf(ta), tf(tb), 4 if max(bm_, bn_) == 256 else 0)
            cls = "big_gemms" if max(bm_, bn_) == 256 else "small_gemm_chain"
            fl = sum(2.0 * p[3] * p[4] * p[5] for p in problems)
            nbytes = sum((p[3] * p[5] + p[5] * p[4]) * 2 + p[3] * p[4] * esz(p[2]) for p in problems)
            return self._timed(cls, name, fl, nbytes, lambda: keep["gemm_grouped"](problems, ta, tb, tile=tile))

        def gemm_grouped_update(problems, upd):
            # all weight gradients + the Adam update of the fusable weights in one launch: the update's 22 B per
            # parameter (theta, m, v read + written, bf16 shadow written) replace the 4 B of the gradient store
            fl = sum(2.0 * p[3] * p[4] * p[5] for p in problems)
            out = keep["gemm_grouped_update"](problems, upd)
            self.fused_numel = sum(hi - lo for lo, hi in out[0])
            return out

        def gemm_grouped_update_timed(problems, upd):
            fused = sum(p[3] * p[4] for p in problems if len(p) > 9 and p[9])
            nbytes = sum((p[3] * p[5] + p[5] * p[4]) * 2 + p[3] * p[4] * 4 for p in problems) + 18.0 * fused
            return self._timed("big_gemms", "k_gemm_grouped256<true, false, false, true, false, true>",
                               sum(2.0 * p[3] * p[4] * p[5] for p in problems), nbytes,
                               lambda: gemm_grouped_update(problems, upd))

        def gemm_kseg(segments, C, M, N, kseg, tb, residual=None):
            name = "k_gemm_kseg<64, 64, 4, %s, 4>" % ("true" if tb else "false")
            n = len(segments)
            return self._timed("small_gemm_chain", name, 2.0 * M * N * kseg * n, n * (M * kseg + kseg * N) * 2 + M * N * 2,
                               lambda: keep["gemm_kseg"](segments, C, M, N, kseg, tb, residual=residual))

        def attn_fwd(q, k, v, out, lse, Bn, nh, Lq, Lk, d, **kw):
            H = nh * d
            return self._timed("attention", None, 4.0 * Bn * Lq * Lk * H, (2 * Bn * Lq + 2 * Bn * Lk) * H * 2,
                               lambda: keep["attn_fwd"](q, k, v, out, lse, Bn, nh, Lq, Lk, d, **kw))

        def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, Bn, nh, Lq, Lk, d, **kw):
            H = nh * d
            fl, by = 10.0 * Bn * Lq * Lk * H, (4 * Bn * Lq + 4 * Bn * Lk) * H * 2
            if kw.get("oproj") is not None:        # the folded o_map dgrad: dY . W_o^T formed inside the launch
                fl += 2.0 * Bn * Lq * H * H
                by += H * H * 2
            return self._timed("attention", None, fl, by,
                               lambda: keep["attn_bwd"](q, k, v, out, dout, lse, dq, dk, dv, Bn, nh, Lq, Lk, d, **kw))

        def add_ln_fwd(x, *a, **kw):
            return self._timed("layernorm", None, 0.0, 4 * x.rows * x.cols * 2, lambda: keep["add_ln_fwd"](x, *a, **kw))

        def add_ln_bwd(dout, *a, **kw):
            return self._timed("layernorm", None, 0.0, 4 * dout.rows * dout.cols * 2,
                               lambda: keep["add_ln_bwd"](dout, *a, **kw))

        def gemm_add_ln(A, Bm, M, N, K, *a, **kw):
            # sub-layer output product + residual + LayerNorm in one launch (zk_gemm_add_ln): the GEMM's operands, the
            # residual read, the sum and the normalised rows written
            bm = 64
            name = "k_gemm_dlds<%d, 64, %d, false, false, 4, %d, 3>" % (bm, 4, 4)
            nbytes = (M * K + K * N) * 2 + 3 * M * N * 2
            return self._timed("small_gemm_chain", name, 2.0 * M * N * K, nbytes,
                               lambda: keep["gemm_add_ln"](A, Bm, M, N, K, *a, **kw))

        def attn_out_ln(q, k, v, att, lse, Bn, nh, Lq, Lk, d, *a, **kw):
            # attention forward + o_map + residual + LayerNorm in one launch (zk_attn_out_ln): booked with the chain (its
            # larger part); falls back to the two launches (booked by their own wrappers) when it returns False
            H = nh * d
            M = Bn * Lq
            fl = 4.0 * Bn * Lq * Lk * H + 2.0 * M * H * H
            by = (2 * Bn * Lq + 2 * Bn * Lk) * H * 2 + (M * H + H * H) * 2 + 3 * M * H * 2
            nkt = (Lk + 63) // 64
            pro = kw["proj"][3] if kw.get("proj") is not None else 0
            if pro:
                # round 6: the merged qkv_map (pro = 3) / q_map (pro = 1) runs inside the launch as well (zk_proj_attn_out_ln):
                # its product is booked here; x and the weight are read, the projected tiles written (the attention tile takes
                # them from the LDS: their read disappears)
                fl += 2.0 * M * H * pro * H
                by += (M * H + H * pro * H) * 2 + M * pro * H * 2 - pro * M * H * 2
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            ok = keep["attn_out_ln"](q, k, v, att, lse, Bn, nh, Lq, Lk, d, *a, **kw)
            e0.record()
            if ok:
                self.records.append(("small_gemm_chain", "k_attn_out_ln<%d, %d>" % (nkt, pro), fl, float(by), s0, e0))
            return ok

        def gemm_ln_bwd(dY, W, M, N, K, *a, **kw):
            # dgrad + the LayerNorm backward its result feeds (zk_gemm_ln_bwd): operands, residual and saved sum read,
            # ds and dy written (the dgrad result itself never leaves the workgroup)
            nbytes = (M * K + K * N) * 2 + 4 * M * N * 2
            return self._timed("small_gemm_chain", "k_gemm_dlds<64, 64, 4, false, true, 4, 4, 4>", 2.0 * M * N * K, nbytes,
                               lambda: keep["gemm_ln_bwd"](dY, W, M, N, K, *a, **kw))

        def gemm_ln(A, Bm, C, M, N, K, bias, np_, **kw):
            # the linear layers of the LayerNorm-free forward (zk_gemm_ln): same tile kernels, LayerNorm in the epilogue
            name = self._name(M, N, K, 0, 0, 0, 0)
            nbytes = (M * K + K * N) * 2 + M * N * 2 + (M * N * 2 if kw.get("residual") is not None else 0)
            return self._timed("small_gemm_chain", name, 2.0 * M * N * K, nbytes,
                               lambda: keep["gemm_ln"](A, Bm, C, M, N, K, bias, np_, **kw))

        def ln_fold(problems):
            nbytes = sum(p[0].numel() * 6 for p in problems)
            return self._timed("layernorm", None, 0.0, nbytes, lambda: keep["ln_fold"](problems))

        def add_ln_bwd_lazy(dout, *a, **kw):
            return self._timed("layernorm", None, 0.0, 5 * dout.rows * dout.cols * 2,
                               lambda: keep["add_ln_bwd_lazy"](dout, *a, **kw))

        def ce_fused(logits, ids, w, ce, dlogits, rows, Vn, ls):
            return self._timed("cross_entropy", None, 0.0, rows * logits.ld * (4 + (2 if dlogits is not None else 0)),
                               lambda: keep["ce_fused"](logits, ids, w, ce, dlogits, rows, Vn, ls))
        for k, fn in (("gemm", gemm), ("gemm_grouped", gemm_grouped), ("gemm_grouped_update", gemm_grouped_update_timed),
                      ("gemm_kseg", gemm_kseg), ("attn_fwd", attn_fwd),
                      ("attn_bwd", attn_bwd), ("add_ln_fwd", add_ln_fwd), ("add_ln_bwd", add_ln_bwd), ("ce_fused", ce_fused),
                      ("gemm_add_ln", gemm_add_ln), ("gemm_ln_bwd", gemm_ln_bwd), ("attn_out_ln", attn_out_ln),
                      ("gemm_ln", gemm_ln), ("ln_fold", ln_fold), ("add_ln_bwd_lazy", add_ln_bwd_lazy)):
            setattr(eng, k, fn)
        if self.top is not None:
            keep["launch_update"] = self.top.launch_update
            numel = self.top.store.numel
            # (the parameters the weight-gradient launch updated itself are not this pass's traffic)
            self.top.launch_update = lambda *a, **kw: self._timed("adam", None, 0.0,
                                                                  30.0 * (numel - getattr(self, "fused_numel", 0)),
                                                                  lambda: keep["launch_update"](*a, **kw))
        return self

    def __exit__(self, *a):
        for k, fn in self._saved.items():
            if k == "launch_update":
                self.top.launch_update = fn
            else:
                setattr(self.eng, k, fn)
        self._saved = {}

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

    def _agg(self, keyfn):
        ovh = self.overhead_s()
        agg = {}
        for rec in self.records:
            key = keyfn(rec)
            if key is None:
                continue
            d = agg.setdefault(key, [0.0, 0.0, 0, 0.0])
            d[0] += rec[2]
            d[1] += max(rec[4].elapsed_time(rec[5]) * 1e-3 - ovh, 1e-7)
            d[2] += 1
            d[3] += rec[3]
        return agg

    def summary(self):
        """{GEMM kernel instance: [flops, seconds, launches, bytes]}; seconds have the event-pair overhead removed."""
        return self._agg(lambda r: r[1])

    def classes(self):
        """{class: [flops, seconds, launches, bytes]}."""
        return self._agg(lambda r: r[0])

    def instances_of(self, cls):
        return self._agg(lambda r: r[1] if r[0] == cls else None)


def current_round():
    """The build round this tree belongs to: one more than the round VERDICT.md reviews (1 without a verdict)."""
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        m = re.search(r"round\s+(\d+)", open(os.path.join(here, "VERDICT.md")).readline())
        return int(m.group(1)) + 1 if m else 1
    except OSError:
        return 1


def tree_stamp():
    """{"round", "commit"} of the code this line was measured on.  The GPU box receives no .git: scripts/gpu.sh writes the
    commit (+dirty) of the snapshot into .head_commit before it travels; in a checkout git answers.  commit null = unknown."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    commit = None
    if os.path.isdir(os.path.join(here, ".git")):
        try:
            commit = subprocess.run(["git", "-C", here, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                                    timeout=10).stdout.strip() or None
            if commit and subprocess.run(["git", "-C", here, "diff", "--quiet"], timeout=10).returncode:
                commit += "+dirty"
        except (OSError, subprocess.SubprocessError):
            commit = None
    else:
        try:
            commit = open(os.path.join(here, ".head_commit")).read().strip() or None
        except OSError:
            pass
    return {"round": current_round(), "commit": commit}


def counter_status(fname, commit):
    """{'stale': bool, 'why': ...} of a committed counter summary: counters come from separate
    rocprofv3 --pmc passes, so a line can only CITE them; it must say when what it cites was measured on other code.  Stale =
    the file belongs to an earlier round (rNN_ prefix below current_round()), carries no commit, or was measured on a
    dirty tree."""
    import re
    if fname is None:
        return None
    m = re.match(r"r(\d+)_", fname)
    rnd = int(m.group(1)) if m else None
    why = []
    if rnd is None or rnd < current_round():
        why.append("measured in round %s, this tree is round %d" % (rnd, current_round()))
    if not commit:
        why.append("no commit recorded")
    elif str(commit).endswith("+dirty"):
        why.append("measured on a dirty tree (%s)" % commit)
    return {"stale": bool(why), "why": "; ".join(why) or None}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.sh on the GPU box in separate
    rocprofv3 --pmc passes -- counters cannot be collected inside the timed run; the file records the commit
    it was measured at); (None, None, None) when no summary holds the kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(f))["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if rec:
            meta = json.load(open(f))
            return (rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"], os.path.basename(f),
                    meta.get("commit"))
    return None, None, None


def pmc_traffic_decode():
    """HBM bytes per decode step from the newest committed summary of scripts/pmc_traffic_decode.sh
    (profiles/*_pmc_traffic_decode.json); (None, None, None) without one."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic_decode.json")), reverse=True):
        try:
            meta = json.load(open(f))
            return meta["bytes_per_step"], os.path.basename(f), meta.get("commit")
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def pmc_mfma(kernel):
    """MFMA-pipe utilisation of `kernel` from the newest committed counter summary (profiles/*_pmc_mfma.json,
    written by scripts/pmc_mfma.sh: one rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and
    GRBM_GUI_ACTIVE over bench.py --steps 2 --no-graph).  mfma_busy = MFMA-busy cycles / (4 SIMDs x 256 CUs x
    kernel cycles); None when no summary holds the kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_mfma.json")), reverse=True):
        try:
            meta = json.load(open(f))
            rec = meta["kernels"].get(kernel)
        except (OSError, ValueError, KeyError):
            continue
        if rec:
            return rec.get("mfma_busy"), os.path.basename(f), meta.get("commit")
    return None, None, None


def cpu_baseline(hp, b=None, budget_s=90.0):
    """Oracle (torch-CPU fp32, unfused, autograd: kind "port") train step on the BENCH batch -- the same
    B=64 x (64+64) synthetic batch the GPU step runs.  Protocol of SURVEY.md 8(d): 3 warm-up + 10 timed steps on
    the host cores when that fits `budget_s` of CPU wall time (judged from the warm-up steps), otherwise as many
    timed steps (>= 1) as fit; the line says which."""
    from oracle import ref_torch as rt
    import copy
    hp = copy.copy(hp)
    src, tgt = synthetic_batch(0, b)
    bs = src.shape[0]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    P = rt.to_torch(rt.init_params(hp, "transformer", seed=1234))
    M = {k: torch.zeros_like(v) for k, v in P.items()}
    Vv = {k: torch.zeros_like(v) for k, v in P.items()}
    feats = {"source": torch.tensor(src), "target": torch.tensor(tgt)}
    # threads: torch's intra-op pool stops scaling on these matrix sizes long before a 256-core host is used up
    # (one step with 256 threads was measured at 425 s against 5.5 s with 64 on the GPU box), so the oracle runs
    # on min(64, cores) threads; `cores` reports that number, `host_cores` what the box has
    cores = max(1, min(64, avail))
    torch.set_num_threads(cores)
    t_begin = time.time()
    step_no, warm, per = 0, 0, None
    while warm < 3:
        t0 = time.time()
        rt.train_step(P, M, Vv, feats, hp, "transformer", step_no, training=True)
        per = time.time() - t0
        step_no += 1
        warm += 1
        if (time.time() - t_begin) + (10 + 3 - warm) * per > budget_s:
            break
    left = budget_s - (time.time() - t_begin)
    want = 10 if left >= 10 * per else max(1, int(left / per))
    n, t0 = 0, time.time()
    while n < want:
        rt.train_step(P, M, Vv, feats, hp, "transformer", step_no, training=True)
        step_no += 1
        n += 1
    dt = time.time() - t0
    return {"value": bs * (LS + LT) * n / dt, "unit": "src+tgt tokens/s", "cores": cores, "host_cores": avail,
            "kind": "port", "warmup_steps": warm, "timed_steps": n, "s_per_step": dt / n,
            "sample": "%d timed steps (+%d warm-up) of the bench batch itself: %d sentences x (64+64) tokens, "
                      "Transformer-base, fwd+bwd+Adam, torch-CPU fp32 restatement of the TF1 path "
                      "(oracle/ref_torch.py; TF1 cannot run here); protocol SURVEY 8(d) = 3+10 when it fits %d s"
                      % (n, warm, bs, int(budget_s))}


# ---------------------------------------------------------------------------------------------
# --mode decode: BASELINE configs[3] (transformer_aan, beam 4, 3000 synthetic sentences, eval batch 32)
# ---------------------------------------------------------------------------------------------
def decode_sources(n, rng):
    """lengths ~ clipped Normal(28, 14) in [4, 100] + eos, ids uniform, length-sorted batches."""
    lens = np.clip(np.rint(rng.normal(28, 14, n)), 4, 100).astype(int)
    order = np.argsort(lens, kind="stable")
    return lens, order


def decode_batch(lens, idx, rng):
    L = int(lens[idx].max()) + 1
    src = np.zeros((len(idx), L), dtype=np.int64)
    for r, i in enumerate(idx):
        src[r, :lens[i]] = rng.integers(3, V, lens[i])
        src[r, lens[i]] = 2
    return src


def decode_step_bytes(hp, model, rows, sent, ls, elem=2):
    """Algorithmic HBM bytes of ONE decode step (SURVEY.md 8(d) last row): every weight the step multiplies
    by, once, as bf16 (elem = 4: the fp32 decode mode reads the fp32 masters and keeps fp32 keys / values); the
    cross-attention K/V of the batch (un-tiled, per sentence); the per-beam caches read and written; the fp32 logits
    written by the GEMM and read by the fused top-k."""
    if elem != 2:
        b2 = decode_step_bytes(hp, model, rows, sent, ls, 2)
        logits = rows * V * 4 * 2
        H, NL = hp.hidden_size, hp.num_decoder_layer
        aan = NL * rows * H * 4 * 2 if model in ("transformer_aan", "transformer_fuse") else 0     # fp32 in both modes
        return (b2 - logits - aan) * elem // 2 + logits + aan
    H, F, NL = hp.hidden_size, hp.filter_size, hp.num_decoder_layer
    per_layer = 2 * H * H + 2 * H * F            # cross q_map, o_map, FFN
    if model == "transformer_aan":
        per_layer += 4 * H * H                   # z_project [2H, 2H]
        cache = rows * H * 4 * 2                 # fp32 running sum read + written
    elif model == "transformer_fuse":
        per_layer += H * H
        cache = rows * H * 4 * 2
    else:
        per_layer += 4 * H * H                   # qkv_map + o_map of the self-attention
        cache = rows * ls * H * 2 * 2            # K/V cache of ~ls positions read (bf16), upper bound
    weights = (NL * per_layer + V * H) * 2
    cross = 2 * NL * sent * ls * H * 2
    logits = rows * V * 4 * 2
    return weights + cross + NL * cache + logits


import threading as _threading
_LANE_FNS = _threading.local()


def decode_measure(args, rank, world):
    """BASELINE configs[3] on this rank's shard; returns the result dict on rank 0 (None elsewhere)."""
    from zero_amd.models import model as registry, load_all
    from zero_amd.search import beam_search
    load_all()
    # BASELINE configs[3] decodes with the average-attention decoder; "transformer_sa" = the plain self-attention model
    model = {"transformer": "transformer_aan", "transformer_sa": "transformer"}.get(args.model, args.model)
    hp = transformer_base_params(model_name=model, scope_name=model, beam_size=4, decode_alpha=0.6,
                                 decode_length=50, eval_batch_size=32)
    hp.src_vocab = SyntheticVocab(V)
    hp.tgt_vocab = SyntheticVocab(V)
    hp.random_seed = 1234
    n_sent = args.sentences
    rng = np.random.default_rng(1234 + rank)         # sentences shard over the ranks with no exchange (main.py:57-62)
    lens, order = decode_sources(n_sent, rng)
    graph = registry.get_model(model)
    enc, dec = graph.infer_fn(hp)
    bsz = hp.eval_batch_size
    batches = [decode_batch(lens, order[b0:b0 + bsz], rng) for b0 in range(0, n_sent, bsz)]
    from zero_amd.evalu import decode_many, decode_streams
    streams = decode_streams() if getattr(args, "decode_streams", 0) <= 0 else args.decode_streams

    def work(src):
        # the thread's own (encoding_fn, decoding_fn): they bind the execution lane of the calling thread
        fns = _LANE_FNS.__dict__.get("fns")
        if fns is None:
            fns = _LANE_FNS.fns = graph.infer_fn(hp)
        return beam_search({"source": src}, fns[0], fns[1], hp)["steps"]
    # warm-up on the LONGEST batches, by EVERY lane: sizes the buffers, pins and kernels of each lane (a lane that meets
    # its first batch inside the ~1.5-s timed region pays ~0.1 s of first-use costs there)
    warm = batches[-max(args.warmup, 1):]
    decode_many(warm, work, streams, each_lane=True)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_counts = decode_many(batches, work, streams)
    steps = sum(step_counts)
    bytes_total = 0.0
    for src, n_steps in zip(batches, step_counts):
        bytes_total += n_steps * decode_step_bytes(hp, model, src.shape[0] * hp.beam_size, src.shape[0],
                                                   int((src != 0).sum(1).mean()))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.cpu()[0])
    if rank != 0:
        return None
    out = {
        "metric": "sentences/sec beam-search decode, %s d=512 L=6, beam 4" % model,
        "value": world * n_sent / dt, "unit": "sentences/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: %s beam=4 alpha=0.6 decode_length=50, %d synthetic sentences per GPU "
                               "(lengths ~ clipped N(28,14) in [4,100] + eos), eval batch 32, length-sorted, cache mode; "
                               "random weights: every batch decodes to its length cap" % (model, n_sent),
                   "rows_per_step": bsz * hp.beam_size, "parallelism": "dp%d (sentences sharded, no exchange)" % world,
                   "batches_in_flight": streams,
                   "hip_graph": "one replay per decode step, search state on the device"},
        "decode_steps_per_s": steps / dt,
        "roofline": {"bound": "hbm", "kernel": "decode step (hipGraph of the whole step: cache reorder + decoder + logits "
                                                "+ fused top-2K + search bookkeeping)",
                     "achieved": bytes_total / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": bytes_total / dt / 8e12, "traffic": pmc_traffic_decode()[0],
                     "traffic_unit": "HBM bytes per decode step (FETCH_SIZE x2 + WRITE_SIZE of every kernel of the job, rocprofv3 PMC)",
                     "traffic_source": pmc_traffic_decode()[1], "traffic_measured_at_commit": pmc_traffic_decode()[2],
                     "algorithmic_bytes_per_step": bytes_total / steps,
                     "note": "achieved = algorithmic bytes of every decode step / wall time of the whole job "
                             "(includes encoder passes and per-batch start-up)"},
    }
    from zero_amd.models._factory import get_core
    out["launches_per_step"] = getattr(get_core(hp, model), "_decode_step_launches", None)
    # the latency of ONE batch's decode step: the same batches one after the other on one lane --
    # a bounded sample (every 8th batch), run once untimed first (buffers, step graphs of these shapes).  `ms_per_step` above is wall time
    # per step with `batches_in_flight` batches overlapping; this is what a single batch waits for a step.
    sample = batches[::8]
    decode_many(sample, work, 1)          # untimed: this lane's buffers and the step graphs of the sample's shapes
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    single_steps = sum(decode_many(sample, work, 1))
    torch.cuda.synchronize()
    out["single_batch_ms_per_step"] = (time.perf_counter() - t1) / max(single_steps, 1) * 1e3
    out["single_batch_sample"] = "%d of the %d batches, one after the other on one lane, %d decode steps" % (
        len(sample), len(batches), single_steps)
    out["parity"] = "bracketed (bf16 product mode; see modes)"
    if model == "transformer_aan" and world == 1 and not getattr(args, "no_modellike", False):
        try:
            ml = out["eos_terminated_weights"] = decode_modellike(args, batches, streams)
            # both decode modes side by side, each with the parity it is held to; the mode that meets
            # the north star's "token-id exact" bar is named first
            out["modes"] = {
                "token_exact_mode": dict(ml["fp32_mode"], mode="decode_dtype=float32 (fp32 masters, activations, accumulation: zk_f32_*)"),
                "throughput_mode": dict(ml["bf16"], mode="decode_dtype=bfloat16 (bf16 shadow weights / activations, fp32 accumulation)"),
                "workload": "the same 3000-sentence job with the EOS-terminating weight set (SURVEY.md 8(d)); `value` above is "
                            "the throughput mode on random weights (every batch runs to its length cap)"}
            out["value_token_exact_mode"] = ml["fp32_mode"]["sentences_per_s"]
        except Exception as exc:      # noqa: BLE001 -- a side measurement must not cost the line
            out["eos_terminated_weights"] = {"error": repr(exc)}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_decode_baseline(hp, model, batches)
    return out


def decode_modellike(args, batches, streams):
    """"... plus a run with a weight set whose eos logit is biased so lengths ~ source lengths".  The same
    3000-sentence job with the weight set of the BASELINE-size decode fixture (tests/fullsize.py beam_params: a random
    Transformer that decodes like a model -- sparse-bigram softmax table, fitted EOS row; built here from
    zero_amd.variables.initial_values, the same numpy stream as the fixture's, without touching oracle/), in the bf16
    product mode with `streams` batches in flight and in the fp32 mode (decode_dtype = float32, zk_f32_*), where the
    hypotheses are token-exact against the fp32 oracle (tests/test_gpu_fullsize.py)."""
    from tests.fullsize import beam_hp, beam_params
    from zero_amd.variables import initial_values
    from zero_amd.models import model as registry
    from zero_amd.models._factory import get_core
    from zero_amd.search import beam_search
    from zero_amd.evalu import decode_many
    res = {"weights": "tests/fullsize.py beam_params (gain-0.1 scope initialiser, scaled target embedding / cross attention, "
                      "successor-map softmax table, fitted EOS row tests/golden/aan_base_beam_eos_row.npy)"}
    src_len = np.concatenate([(b != 0).sum(1) for b in batches])
    for tag, dd, lanes in (("bf16", "bfloat16", streams), ("fp32_mode", "float32", streams)):
        hp = beam_hp()
        hp.decode_dtype = dd
        hp.scope_name = "bench_modellike_" + tag
        model = hp.model_name
        Pn = beam_params(hp, model, init=lambda hp_, m_, seed_: initial_values(hp_, m_, seed_))
        get_core(hp, model, Pn)
        graph = registry.get_model(model)
        key = "fns_" + tag

        def work(src, hp=hp, key=key, graph=graph):
            fns = _LANE_FNS.__dict__.get(key)
            if fns is None:
                fns = graph.infer_fn(hp)
                setattr(_LANE_FNS, key, fns)
            r = beam_search({"source": src}, fns[0], fns[1], hp)
            best = np.asarray(r["seq"])[:, 0]
            return r["steps"], [int((row == 2).any()) for row in best], \
                [int(np.argmax(row == 2)) + 1 if (row == 2).any() else int((row != 0).sum()) for row in best]
        job = batches if tag == "bf16" else batches[::2]          # the fp32 mode on half of the job (bounded)
        decode_many(job[-max(args.warmup, 1):], work, lanes, each_lane=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = decode_many(job, work, lanes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = sum(g[0] for g in got)
        ends = sum(sum(g[1]) for g in got)
        lens = np.concatenate([np.asarray(g[2]) for g in got])
        n = int(sum(b.shape[0] for b in job))
        elem = 2 if tag == "bf16" else 4
        nbytes = sum(g[0] * decode_step_bytes(hp, model, b.shape[0] * hp.beam_size, b.shape[0], int((b != 0).sum(1).mean()), elem)
                     for b, g in zip(job, got))
        # one batch's step latency: a bounded sample of the job on ONE lane (run once untimed: buffers, step graphs)
        sample = job[::8]
        decode_many(sample, work, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s_got = decode_many(sample, work, 1)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        s_steps = sum(g[0] for g in s_got)
        s_bytes = sum(g[0] * decode_step_bytes(hp, model, b.shape[0] * hp.beam_size, b.shape[0], int((b != 0).sum(1).mean()), elem)
                      for b, g in zip(sample, s_got))
        res[tag] = {"sentences_per_s": n / dt, "sentences": n, "decode_steps": steps, "ms_per_step": dt / steps * 1e3,
                    "batches_in_flight": lanes, "eos_terminated_frac": ends / float(n), "mean_hypothesis_len": float(lens.mean()),
                    "mean_source_len": float(src_len.mean()), "decode_dtype": dd,
                    "parity": PARITY[tag],
                    "single_batch_ms_per_step": dt1 / max(s_steps, 1) * 1e3,
                    "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s", "bytes": "algorithmic, %d-byte weights / keys / values" % elem,
                                 "achieved": nbytes / dt / 1e9, "frac": nbytes / dt / 8e12,
                                 "single_batch_achieved": s_bytes / dt1 / 1e9, "single_batch_frac": s_bytes / dt1 / 8e12}}
    return res


# what tests/test_gpu_fullsize.py holds each decode mode to at d = 512, V = 32000 (256 sentences, beam 1 and 4)
PARITY = {"fp32_mode": "token-exact: 256 / 256 best hypotheses (and every beam) equal to the fp32 oracle's, beam 1 and 4, scores "
                       "within 1e-4 relative (test_aan_beam_search_base_size_fp32_is_token_exact) -- the north star's bar",
          "bf16": "bracketed: 209 (beam 1) / 203 (beam 4) of 256 token-exact against the fp32 oracle, where the bf16-storage "
                  "ORACLE reaches 206 / 190 (and a second bf16-storage oracle that differs only in the order of its fp32 "
                  "sums: 207 / 197, agreeing with the first on 214 / 198 -- profiles/r06_bf16_oracle_noise_floor.txt); every "
                  "first divergence a near-tie of the fp32 oracle (test_aan_beam_search_base_size)"}


def decode_main(args, rank, world):
    out = decode_measure(args, rank, world)
    if out is not None:
        print(json.dumps(out))


def cpu_decode_baseline(hp, model, batches):
    """Oracle beam_search (oracle/ref_torch.py restatement of search.py:19-275, kind "port") on a bounded sample of
    the same workload: the first 8 sentences of a middle batch."""
    from oracle import ref_torch as rt
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(64, avail))
    torch.set_num_threads(cores)
    P = rt.to_torch(rt.init_params(hp, model, seed=1234))
    enc, dec = rt.infer_fn(hp, P, model)
    src = batches[len(batches) // 2][:8]
    t0 = time.time()
    r = rt.beam_search({"source": torch.tensor(src)}, enc, dec, hp)
    dt = time.time() - t0
    return {"value": src.shape[0] / dt, "unit": "sentences/s", "cores": cores, "kind": "port",
            "decode_steps_per_s": r["steps"] / dt,
            "sample": "8 sentences (one eval batch is 32) of the middle length bucket, beam 4, %d decode steps, "
                      "torch-CPU fp32 restatement of search.py + transformer_aan.py" % r["steps"]}


def choose_headline(legs):
    """The leg whose time is the line's `value`: the fastest REFERENCE-EXACT leg (fp32 gradient buckets -- what
    utils/parallel.py:184-196 averages) unless a bf16-bucket leg beats it by more than 3 %.  legs: dicts with `ms_per_step`
    and `reference_exact` (skipped legs carry neither and are ignored); None if no fp32 leg finished."""
    done = [l for l in legs if "ms_per_step" in l]
    exact = [l for l in done if l.get("reference_exact")]
    if not exact:
        return None
    best32 = min(exact, key=lambda l: l["ms_per_step"])
    best16 = min((l for l in done if not l.get("reference_exact")), key=lambda l: l["ms_per_step"], default=None)
    return best16 if (best16 is not None and best16["ms_per_step"] < 0.97 * best32["ms_per_step"]) else best32


def spawn_ranks(n):
    """Re-launch this command line under `python -m torch.distributed.run --standalone --nproc-per-node n`."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--mode", choices=["train", "decode"], default="train",
                    help="train = the headline metric (default); decode = BASELINE configs[3] (beam-4 decode, sentences/s)")
    ap.add_argument("--sentences", type=int, default=3000, help="decode leg: sentences per GPU")
    ap.add_argument("--sentences-per-gpu", type=int, default=B,
                    help="training batch in sentences per GPU (default 64 = the metric's 4096+4096 tokens; other "
                         "values are SIDE measurements that separate kernel quality from 'problem too small for 256 CUs')")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode leg of the default line")
    ap.add_argument("--static-batch", action="store_true",
                    help="SIDE measurement: time the replay of ONE pre-uploaded batch (rounds 1-3's loop) instead of "
                         "Trainer.step on rotating batches; the default line carries both")
    ap.add_argument("--no-modellike", action="store_true",
                    help="decode leg: skip the side run with the EOS-terminating weight set (bf16 and fp32 mode)")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up + the timed loop and nothing else (no second loop, no instrumented pass, no decode / CPU "
                         "legs): what a rocprofv3 --kernel-trace of the CAPTURED step should see (scripts/gpu_round5.sh profstep)")
    ap.add_argument("--decode-streams", type=int, default=0,
                    help="decode batches in flight at once (0 = ZERO_HIP_DECODE_STREAMS or its default)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: spawn the N ranks ourselves (one process per GPU, torchrun's
        # standalone rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  The torchrun form of the contract
        # (WORLD_SIZE set by the launcher) goes straight on below.
        sys.exit(spawn_ranks(args.gpus))
    rank, world, local = parallel.init_distributed()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)

    from zero_amd.main import Trainer
    from zero_amd import hip as _hip
    for kv in os.environ.get("ZERO_HIP_TUNE", "").split(","):     # e.g. ZERO_HIP_TUNE=0:0 (A/B switches)
        if ":" in kv:
            _hip.lib().raw("zk_tune")(int(kv.split(":")[0]), int(kv.split(":")[1], 0))
    if args.mode == "decode":
        decode_main(args, rank, world)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    hp = make_params(args.dropout, args.size, args.model)
    hp.random_seed = 1234   # identical initial replicas on every rank
    tr = Trainer(hp)
    nb = args.sentences_per_gpu
    feats = [dict(zip(("source", "target"), synthetic_batch(rank, nb, i))) for i in range(ROTATION)]
    tr.core.eng.set_seed(1234 + rank)
    # one rank: the whole step is one hipGraph; several ranks: hipGraph segments between the
    # gradient-bucket hand-offs to RCCL (Trainer._step_segmented)
    use_graph = not args.no_graph

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, steps, warm):
        """warm untimed steps, then EXACTLY `steps` steps between barrier + synchronize; max over ranks (seconds)."""
        loss = None
        import contextlib
        # as zero_amd.main.train issues its steps: from the engine's work stream (ZERO_HIP_BENCH_DEFAULT_STREAM=1: from the
        # default stream, with a stream hand-off per step -- the A/B of profiles/r05_rocprof_captured_step_gaps*.txt)
        with (contextlib.nullcontext() if os.environ.get("ZERO_HIP_BENCH_DEFAULT_STREAM") == "1" else tr.on_work_stream()):
            for i in range(warm):
                step_fn(i)
            barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                loss = step_fn(warm + i)
            barrier()
            dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        if world > 1:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.cpu()[0]), loss

    def rotating(i):          # the loop a user runs: a NEW batch every step (upload + id prep + step)
        return tr.step(feats[i % ROTATION], use_graph)

    def static(i):            # one pre-uploaded batch replayed (rounds 1-3's measurement: the device work alone)
        return tr.step_static(use_graph)

    # ---- multi-GPU: one short timed leg per exchange mode, in this process group
    legs, chosen = [], None
    ranks_seen_box = [None]

    def gather_ranks_seen():
        # which physical devices the ranks sit on (all-gather of the device UUIDs): SCALE shows N distinct GPUs
        try:
            me = str(torch.cuda.get_device_properties(local).uuid)
        except Exception:      # noqa: BLE001
            me = "%s:%d" % (os.uname().nodename, local)
        seen = [None] * world
        torch.distributed.all_gather_object(seen, me)
        return {"distinct_devices": len(set(seen)), "device_uuids": seen}
    if world > 1:
        import threading

        def bail(name):
            # a leg that does not come back (a collective of the optional direct transport waiting for a rank that never
            # arrives) must not cost the run its line: print what was measured so far and leave
            if rank == 0:
                head = choose_headline(legs) or chosen       # the same rule as a complete run, over the legs that finished
                line = bench_line(head, legs, aborted=name)
                line.setdefault("rccl", {})["ranks_seen"] = ranks_seen_box[0]
                print(json.dumps(line), flush=True)
            os._exit(0)

        def run_leg(dtype, direct, sparse, guard_s=None):
            name = "%s/%s/%s" % (dtype, "zk_comm" if direct else "torch", "rows" if sparse else "dense")
            if os.environ.get("ZERO_HIP_BENCH_FAKE_HANG") == name and not guard_s:      # test hook: guard the leg that will hang
                guard_s = float(os.environ.get("ZERO_HIP_BENCH_GUARD_S", "120"))
            timer = None
            if guard_s:
                timer = threading.Timer(guard_s, bail, args=(name,))
                timer.daemon = True
                timer.start()
            try:
                if os.environ.get("ZERO_HIP_BENCH_FAKE_HANG") == name:     # test hook: a leg that never comes back
                    time.sleep(3600)
                ok = parallel.select_transport(direct)
                if direct and not ok:
                    legs.append({"leg": name, "skipped": "the direct communicator did not come up on every rank"})
                    return None
                tr.reducer.configure(bucket_dtype=dtype, sparse=sparse)
                dt, loss = timed(rotating, args.steps, 3)
                leg = {"leg": name, "bucket_dtype": dtype, "transport": "zk_comm" if direct else
                       "torch.distributed:%s" % torch.distributed.get_backend(), "sparse_rows_exchange": tr.reducer.sparse_keys(),
                       "ms_per_step": dt / args.steps * 1e3, "bytes_per_rank_per_step": tr.reducer.bytes_last_step,
                       "reference_exact": dtype == "fp32", "_dt": dt, "_loss": loss}
                legs.append(leg)
                return leg
            finally:
                if timer is not None:
                    timer.cancel()

    def bench_line(head, legs, aborted=None):
        """The JSON object of the run so far (the multi-GPU watchdog prints it early if a later leg hangs)."""
        dt = head["_dt"]
        ms = dt / args.steps * 1e3
        flops = train_flops_per_step(hp, b=nb)
        out = {
            "metric": "src+tgt tokens/sec training, Transformer-%s d=%d L=6" % (args.size, hp.hidden_size),
            "value": world * nb * (LS + LT) * args.steps / dt, "unit": "src+tgt tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"model_name": args.model,
                       "workload": "Transformer-%s (d=%d, L=6+6, F=%d, h=%d, V=32000) training step, "
                                   "B=%d x (src 64 + tgt 64) tokens per GPU, dropout %.2f, label_smooth 0.1, "
                                   "fwd+bwd+allreduce+Adam%s%s" % (args.size, hp.hidden_size, hp.filter_size,
                                                                   hp.num_heads, nb, args.dropout,
                                                                   "" if nb == B else " [SIDE MEASUREMENT: not the metric's batch]",
                                                                   " [SIDE MEASUREMENT: one static batch replayed]" if args.static_batch else ""),
                       "global_batch_tokens": world * nb * (LS + LT), "parallelism": "dp%d" % world,
                       "timed_loop": "one pre-uploaded batch replayed (Trainer.step_static)" if args.static_batch else
                                     "Trainer.step(features) on a rotation of %d distinct batches per GPU: pinned async "
                                     "id upload + device-side id prep (masks, loss weights, rows grouped by id) + step"
                                     % ROTATION,
                       "hip_graph": ("whole step" if world == 1 else "segments between all-reduce buckets") if use_graph else False},
            "step_mfma_frac": flops / (ms * 1e-3) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
        }
        if legs:
            out["rccl"] = {"ranks": world, "legs": [{k: v for k, v in l.items() if not k.startswith("_")} for l in legs],
                           "headline_leg": head.get("leg"), "aborted_leg": aborted}
        return out

    if world == 1:
        if args.static_batch:
            tr.prepare_static(feats[0])
        dt, loss = timed(static if args.static_batch else rotating, args.steps, max(args.warmup, 2))   # >= 2: eager sizing pass + capture
        chosen = {"_dt": dt, "_loss": loss}
    else:
        # reference-exact first (fp32 buckets, torch.distributed = RCCL): the line exists before anything optional runs
        chosen = run_leg("fp32", False, False)
        ranks_seen_box[0] = gather_ranks_seen()      # before anything optional runs: a leg that hangs later cannot lose it
        run_leg("fp32", False, True)
        run_leg("bf16", False, True)
        run_leg("bf16", False, False)
        if os.environ.get("ZERO_HIP_BENCH_DIRECT", "1") != "0":
            guard = float(os.environ.get("ZERO_HIP_BENCH_GUARD_S", "120"))
            run_leg("fp32", True, True, guard_s=guard)
            run_leg("bf16", True, True, guard_s=guard)
        # headline = the fastest reference-exact (fp32) leg unless bf16 buckets win by more than 3 %
        chosen = choose_headline(legs)
        # leave the reducer in the chosen mode for what follows
        parallel.select_transport(chosen["transport"] == "zk_comm")
        tr.reducer.configure(bucket_dtype=chosen["bucket_dtype"], sparse=bool(chosen["sparse_rows_exchange"]))
        for _ in range(2):
            rotating(0)
    dt, loss = chosen["_dt"], chosen["_loss"]
    if args.timed_only:
        if rank == 0:
            line = bench_line(chosen, legs)
            line["timed_only"] = True
            line["launches_per_step"] = getattr(tr.core.eng, "last_graph_nodes", None) if (use_graph and world == 1) else None
            print(json.dumps(line))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    loss_v = float(loss.reshape(-1)[0].cpu())
    gnorm, pnorm, skipped = tr.train_op.stats()
    step_launches = getattr(tr.core.eng, "last_graph_nodes", None) if (use_graph and world == 1) else None

    # ---- the other timed loop of the same line (one rank): static batch beside rotating batches
    ms_other = None
    if world == 1:
        tr.prepare_static(feats[0])
        d2, _ = timed(rotating if args.static_batch else static, args.steps, 2)
        ms_other = d2 / args.steps * 1e3

    # ---- what the exchange costs: the same K steps with the gradient buckets NOT handed to RCCL (every rank updates
    # from its local gradients; the replicas drift apart, which is harmless after the timed region).  exposed =
    # step time - this.  Runs the multi-rank path without collectives, so it is an upper bound of the one-rank step.
    ms_nocomm = None
    if world > 1:
        tr.reducer.disabled = True
        dn, _ = timed(rotating, args.steps, 2)
        ms_nocomm = dn / args.steps * 1e3
        tr.reducer.disabled = False

    # ---- roofline: instrumented eager pass (HIP events per launch).  Every
    # rank runs it (the step contains collectives); only rank 0 reports.
    # Each pass is enqueued behind a 15 ms spin kernel on the launch stream, so the launches are
    # already queued when they run and the brackets do not contain the host launch latency.
    NPROF = 3
    eng = tr.core.eng
    tr.prepare_static(feats[0])
    with LaunchProfiler(eng, tr.train_op) as prof:
        for _ in range(NPROF):
            torch.cuda.synchronize()
            eng.lib.call("zk_spin", 15000, eng.work_stream.cuda_stream)
            tr.step_static(False)
        torch.cuda.synchronize()
        with torch.cuda.stream(eng.work_stream):
            eng.lib.call("zk_spin", 15000, eng.work_stream.cuda_stream)   # long enough for the host to enqueue all of it
            prof.calibrate()
    agg = prof.summary()
    cls_agg = prof.classes()
    barrier()
    ranks_seen = ranks_seen_box[0]
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    ms = dt / args.steps * 1e3
    flops = train_flops_per_step(hp, b=nb)
    out = bench_line(chosen, legs)
    out.update({"loss": loss_v, "gnorm": gnorm, "update_skipped": skipped, "launches_per_step": step_launches})
    out["tree"] = tree_stamp()
    # the in-launch LayerNorm exchange bounds every wait and records a give-up on the device: a line measured with one is void
    out["sync_ln_errors"] = int(tr.core.eng.sync_ln_errors())
    if out["sync_ln_errors"]:
        raise RuntimeError("bench.py: a workgroup of an in-launch LayerNorm exchange gave up waiting for its peers; "
                           "the measurement is void (ZERO_HIP_SYNC_LN=0 runs the launch structure without it)")
    if ms_other is not None:
        out["static_batch_ms_per_step" if not args.static_batch else "rotating_batches_ms_per_step"] = ms_other
        rot, sta = (ms, ms_other) if not args.static_batch else (ms_other, ms)
        out["feed_overhead_frac"] = rot / sta - 1.0          # what the per-batch upload + id prep add to the replayed step
    tp = parallel.transport()
    rc = out.setdefault("rccl", {"ranks": world, "legs": [], "headline_leg": None, "aborted_leg": None})
    rc.update({"transport": None if world == 1 else ("zk_comm" if tp is not None else "torch.distributed:%s"
                                                     % torch.distributed.get_backend()),
               "bucket_dtype": tr.reducer.bucket_dtype_name() if world > 1 else None,
               "sparse_rows_exchange": tr.reducer.sparse_keys() if world > 1 else None,
               "bytes_per_rank_per_step": chosen.get("bytes_per_rank_per_step", 0) if world > 1 else 0,
               "ms_per_step_without_exchange": ms_nocomm,
               "exposed_allreduce_ms": (ms - ms_nocomm) if ms_nocomm is not None else 0.0,
               "ranks_seen": ranks_seen})
    for leg in rc["legs"]:
        if ms_nocomm is not None and "ms_per_step" in leg:
            leg["exposed_ms"] = leg["ms_per_step"] - ms_nocomm
    # ---- kernel classes (roofline.by_class): the headline is the class with the largest time per step, so that it cannot
    # move by splitting or merging template instances; `other` = the step minus every booked class (embeddings, masks /
    # id prep, loss, column / LayerNorm-parameter reductions, zero fill, and the gaps between graph nodes)
    by_class, booked_us = {}, 0.0
    for cname, bound, what in CLASSES:
        if cname not in cls_agg:
            continue
        fl, sec, cnt, by = cls_agg[cname]
        us = sec / NPROF * 1e6
        booked_us += us
        ent = {"bound": bound, "launches_per_step": cnt // NPROF, "us_per_step": us, "what": what}
        if bound == "mfma":
            ent.update({"achieved": fl / sec / 1e12, "unit": "TFLOP/s", "peak": MFMA_BF16_PEAK_TFLOPS,
                        "frac": fl / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS, "gflop_per_step": fl / NPROF / 1e9})
        else:
            ent.update({"achieved": by / sec / 1e12, "unit": "TB/s", "peak": HBM_PEAK_TBS,
                        "frac": by / sec / 1e12 / HBM_PEAK_TBS, "algorithmic_mb_per_step": by / NPROF / 1e6})
            if fl:
                ent["tflops"] = fl / sec / 1e12
        by_class[cname] = ent
    by_class["other"] = {"bound": "latency", "us_per_step": max(ms * 1e3 - booked_us, 0.0),
                         "what": "step time minus the booked classes: embeddings, id prep, loss, column / LayerNorm-"
                                 "parameter reductions, zero fill, gaps between graph nodes"}
    top_cls = max((c for c in by_class if c != "other"), key=lambda c: by_class[c]["us_per_step"])
    inst = prof.instances_of(top_cls)
    # worst instance of the dominant class = the one with the largest total time
    ranked = sorted(inst, key=lambda k: -inst[k][1]) if inst else []
    ranked_all = sorted(agg, key=lambda k: -agg[k][1])
    key = ranked[0] if ranked else (ranked_all[0] if ranked_all else None)
    traffic, traffic_src, traffic_commit = pmc_traffic(key)
    mfma_busy, mfma_src, mfma_commit = pmc_mfma(key)
    top = by_class[top_cls]
    tot_fl = sum(v[0] for v in agg.values())
    tot_s = sum(v[1] for v in agg.values())
    out["roofline"] = {
        "bound": top["bound"], "class": top_cls, "kernel": "%s: %s" % (top_cls, top["what"]),
        "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
        "total_us_per_step": top["us_per_step"], "launches_per_step": top["launches_per_step"],
        "worst_instance": ({"kernel": key, "frac": agg[key][0] / agg[key][1] / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                            "tflops": agg[key][0] / agg[key][1] / 1e12,
                            "total_us_per_step": agg[key][1] / NPROF * 1e6, "launches_per_step": agg[key][2] // NPROF,
                            "avg_launch_us": agg[key][1] / agg[key][2] * 1e6,
                            "flop_per_launch": agg[key][0] / agg[key][2]} if key in agg else None),
        "traffic": traffic, "traffic_unit": "bytes/launch of worst_instance (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 PMC)",
        "traffic_source": traffic_src, "traffic_measured_at_commit": traffic_commit,
        "traffic_status": counter_status(traffic_src, traffic_commit),
        "mfma_busy": mfma_busy, "mfma_busy_source": mfma_src, "mfma_busy_measured_at_commit": mfma_commit,
        "mfma_busy_status": counter_status(mfma_src, mfma_commit),
        "step_frac": flops / (ms * 1e-3) / (MFMA_BF16_PEAK_TFLOPS * 1e12),
        "event_pair_overhead_us": prof.overhead_s() * 1e6,
        "all_gemm_achieved": tot_fl / tot_s / 1e12, "all_gemm_ms_per_step": tot_s / NPROF * 1e3,
        "by_class": by_class,
        "by_kernel": {k: {"launches_per_step": v[2] // NPROF, "avg_us": v[1] / v[2] * 1e6,
                          "tflops": v[0] / v[1] / 1e12} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])},
    }
    if world == 1 and not args.no_decode and args.model == "transformer" and args.size == "base" and nb == B:
        # BASELINE configs[3] under the same clock: its own model / parameter store, after the training leg
        try:
            dargs = argparse.Namespace(**vars(args))
            dargs.warmup = max(args.warmup, 1)      # per lane, on the longest batches (untimed): buffers, pins, kernels, graph cache
            out["decode"] = decode_measure(dargs, rank, world)
        except Exception as exc:      # noqa: BLE001 -- the headline line must still be printed
            out["decode"] = {"error": repr(exc)}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(hp, nb)     # bounded sample, see cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
