# coding: utf-8
"""Full-size oracle fixtures (run in the BUILD container only).

The HIP path is otherwise only compared with the oracle at toy sizes; the shapes that select the
benchmark's kernels (64x64 producer-wave tiles, the 12-segment K-segmented GEMM, the grouped weight
gradients incl. the logits problem, split-K) need a comparison at the BASELINE sizes.  The oracle
(oracle/ref_torch.py, fp32 torch-CPU restatement; PARITY UNPINNED -- TF1 never ran) takes minutes at
these sizes, so its outputs are committed here and the `-m gpu` tests compare against them:

  base_synth_seed1234.npz  BASELINE configs[1]: d=512, F=2048, h=8, 6+6 layers, V=32000, B=64 x (64+64),
                           dropout 0, label smoothing 0.1
  enc12_synth_seed1234.npz BASELINE configs[4], the part the reference has code for: the same model with
                           num_encoder_layer=12 (transformer.py:35-69)
  big_synth_seed1234.npz   BASELINE configs[2] widths: d=1024, F=4096, h=16, 6+6 layers
  aan_base_beam.npz        BASELINE configs[3] subset: transformer_aan, d=512, 256 sentences, beam 1 and 4, by the fp32
                           oracle and by the oracle under the bf16 storage model (keys bf16_*).  Round 5: the weight set
                           of tests/fullsize.py beam_params (decodes like a model: hypotheses end in EOS at lengths around
                           the source length, few repeats; `stats_k*` = STAT_KEYS below)

Parameters are NOT stored (77-242 M floats): both sides regenerate them from
``oracle.ref_torch.init_params(hp, model, seed)`` + ``tests.common.perturb`` (numpy Generator streams are
stable across platforms); `param_probe` pins that the regenerated values are the ones used here.

Each training fixture holds, for the fp32 oracle and for the oracle under the bf16 storage model
(``Cfg.store_bf16``: tensors the HIP path keeps as bf16 are rounded at the same points, arithmetic fp32):
loss, per-sentence loss, global gradient norm, the L2 norm of every variable's gradient, and three
gradient slices.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_torch as rt  # noqa: E402
from tests.common import perturb  # noqa: E402
from tests.fullsize import (fullsize_hp, fullsize_batch, fullsize_params, param_probe, SLICES,  # noqa: E402
                            beam_hp, beam_params, beam_sources, BEAM_SENTENCES)


def train_fixture(name, **kw):
    hp = fullsize_hp(**kw)
    model = hp.model_name
    Pn = fullsize_params(hp, model)
    src, tgt = fullsize_batch()
    out = {"param_probe": param_probe(Pn)}
    for tag, flag in (("f32", False), ("bf16", True)):
        rt.Cfg.store_bf16 = flag
        t0 = time.time()
        P = rt.to_torch(Pn, torch.float32, requires_grad=True)
        r = rt.train_fn({"source": torch.tensor(src), "target": torch.tensor(tgt)}, hp, P, model, training=False)
        r["loss"].backward()
        rt.Cfg.store_bf16 = False
        names = list(P.keys())
        gn = np.array([float(P[k].grad.double().norm()) if P[k].grad is not None else 0.0 for k in names])
        out[tag + "_loss"] = np.float64(r["loss"].item())
        out[tag + "_per_sample"] = r["per_sample_loss"].detach().numpy().astype(np.float32)
        out[tag + "_gnorm"] = np.float64(np.sqrt((gn ** 2).sum()))
        out[tag + "_grad_norms"] = gn
        for i, (k, rs, cs) in enumerate(SLICES):
            out["%s_slice%d" % (tag, i)] = P[k].grad[rs[0]:rs[1], cs[0]:cs[1]].numpy().astype(np.float32)
        print("%s %s: loss %.6f gnorm %.6f (%.0f s)" % (name, tag, out[tag + "_loss"], out[tag + "_gnorm"],
                                                       time.time() - t0), flush=True)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


STAT_KEYS = ("eos_terminated_frac", "repeat_frac", "mean_len", "mean_src_len", "len_minus_src_std", "corr_len_src",
             "min_boundary_gap", "p1_boundary_gap", "median_boundary_gap", "decode_steps")


def beam_stats(best, src, tsc, K):
    """What makes the fixture a decode WORKLOAD: share of best hypotheses that end in an EOS, share
    of positions that repeat the previous token, hypothesis lengths against source lengths, and the score gaps at the
    boundary that decides which candidates stay alive (rank K against K + 1 of the step's table)."""
    slen = (src != 0).sum(1)
    ends, lens, reps, tot = 0, [], 0, 0
    for s in best:
        s = [int(x) for x in s]
        if 2 in s:
            ends += 1
            L = s.index(2) + 1
        else:
            L = len([x for x in s if x != 0])
        lens.append(L)
        reps += sum(1 for a, b in zip(s[1:L], s[:L - 1]) if a == b)
        tot += max(L - 1, 1)
    lens = np.array(lens, dtype=np.float64)
    with np.errstate(invalid="ignore"):
        g = tsc[:, :, K - 1] - tsc[:, :, K]
    g = g[np.isfinite(g) & (tsc[:, :, K] > -1e30)]
    return {"eos_terminated_frac": ends / float(len(best)), "repeat_frac": reps / float(tot), "mean_len": lens.mean(),
            "mean_src_len": slen.mean(), "len_minus_src_std": (lens - slen).std(),
            "corr_len_src": float(np.corrcoef(lens, slen)[0, 1]), "min_boundary_gap": g.min(),
            "p1_boundary_gap": np.percentile(g, 1), "median_boundary_gap": np.median(g),
            "decode_steps": float(np.isfinite(tsc[:, :, 0]).any(axis=1).sum())}


def beam_fixture():
    """256 sentences (round 4; 64 before), each decoded by the fp32 oracle AND by the oracle under the bf16 storage
    model (Cfg.store_bf16: the tensors the HIP path keeps as bf16 rounded at the same points).  The second run is what
    turns a greedy miss from "explained" (a near-tie of the fp32 oracle) into "reproduced": at the step where the HIP
    search leaves the fp32 oracle's path, tests/test_gpu_fullsize.py checks whether the bf16-storage oracle makes the
    same choice as the HIP path, or whether the gap is inside the two oracles' own disagreement at that step."""
    hp = beam_hp()
    model = hp.model_name
    Pn = beam_params(hp, model)          # round 5: the weight set that decodes like a model (tests/fullsize.py)
    P = rt.to_torch(Pn)
    src = beam_sources(BEAM_SENTENCES)
    out = {"param_probe": param_probe(Pn), "source": src}
    enc, dec = rt.infer_fn(hp, P, model)
    for prefix, flag in (("", False), ("bf16_", True)):
        for K in (1, 4):
            hp.beam_size = K
            t0 = time.time()
            seqs, scores, traces = [], [], []
            for i in range(0, src.shape[0], 32):
                hp.search_trace = []
                rt.Cfg.store_bf16 = flag
                try:
                    r = rt.beam_search({"source": torch.tensor(src[i:i + 32])}, enc, dec, hp)
                finally:
                    rt.Cfg.store_bf16 = False
                seqs.append(np.asarray(r["seq"])); scores.append(np.asarray(r["score"]))
                traces.append(hp.search_trace)
                hp.search_trace = None
            # per-step candidate tables of the oracle: the 2K candidates search.py:172-176 keeps + the runner-up (scores
            # fp32, flat index beam * V + token), [steps, sentences, 2K+1], steps past a batch's end are NaN / -1.
            # tests/test_gpu_fullsize.py uses them to measure the score gap at the step where the HIP search first
            # leaves the oracle's path.
            T = max(len(t) for t in traces)
            W = 2 * K + rt.TRACE_RUNNER_UPS
            tsc = np.full((T, src.shape[0], W), np.nan, dtype=np.float32)
            tix = np.full((T, src.shape[0], W), -1, dtype=np.int32)
            for bi, tr in enumerate(traces):
                for t, (a, b) in enumerate(tr):
                    tsc[t, bi * 32:bi * 32 + a.shape[0], :a.shape[1]] = a
                    tix[t, bi * 32:bi * 32 + a.shape[0], :a.shape[1]] = b
            out[prefix + "trace_scores_k%d" % K] = tsc
            out[prefix + "trace_idx_k%d" % K] = tix
            L = max(s.shape[-1] for s in seqs)
            seqs = [np.pad(s, [(0, 0)] * (s.ndim - 1) + [(0, L - s.shape[-1])]) for s in seqs]
            out[prefix + "seqs_k%d" % K] = np.concatenate(seqs, 0).astype(np.int32)
            out[prefix + "scores_k%d" % K] = np.concatenate(scores, 0).astype(np.float32)
            print("%sbeam %d: %s (%.0f s)" % (prefix, K, out[prefix + "seqs_k%d" % K].shape, time.time() - t0), flush=True)
            st = beam_stats(out[prefix + "seqs_k%d" % K][:, 0], src, tsc, K)
            out[prefix + "stats_k%d" % K] = np.array([st[k] for k in STAT_KEYS], dtype=np.float64)
            print("   " + ", ".join("%s %.4g" % (k, st[k]) for k in STAT_KEYS), flush=True)
            np.savez_compressed(os.path.join(HERE, "aan_base_beam.partial.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "aan_base_beam.npz"), **out)
    os.remove(os.path.join(HERE, "aan_base_beam.partial.npz"))


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    which = sys.argv[1:] or ["base", "enc12", "big", "beam"]
    if "base" in which:
        train_fixture("base_synth_seed1234")
    if "enc12" in which:
        train_fixture("enc12_synth_seed1234", num_encoder_layer=12)
    if "big" in which:
        train_fixture("big_synth_seed1234", hidden_size=1024, embed_size=1024, filter_size=4096, num_heads=16)
    if "beam" in which:
        beam_fixture()
