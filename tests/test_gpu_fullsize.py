"""HIP path against the oracle at the BASELINE sizes: the shapes that select
the benchmark's kernels -- 64x64 producer-wave tiles, 128/256-wide tiles of the logits trio, the grouped weight
gradients, the 12-segment K-segmented GEMM, split-K -- are only reached here.

The oracle's outputs come from committed fixtures (tests/golden/*_synth_seed1234.npz, aan_base_beam.npz;
generated in the build container by tests/golden/make_fullsize_golden.py from oracle/ref_torch.py -- PARITY
UNPINNED, TF1 never ran); parameters and ids are regenerated on both sides from the same numpy streams.

Tolerances:
  loss            within 1e-3 relative of the fp32 oracle (north-star tolerance);
  per-sentence    within 2e-3 relative;
  gradients       compared with the oracle run under the bf16 STORAGE model (Cfg.store_bf16: every tensor the
                  HIP path keeps as bf16 is rounded at the same point, arithmetic fp32), so that what is left is
                  accumulation order and the few roundings the model does not place identically:
                  global norm within 1e-2, EVERY variable's gradient norm within 1.5e-2 (measured <= 1.1 %, also
                  for the variables whose gradient is 1e-4 of the largest), slices within SLICE_TOL relative L2;
                  against the pure fp32 oracle the global norm must stay within 3e-2;
  beam search     the weight set that decodes like a model (tests/fullsize.py beam_params): every oracle hypothesis ends
                  in an EOS at a length around its source's, 5-10 % of the positions repeat, the two ORACLES (fp32 / bf16
                  storage model) agree with each other on 206 (beam 1) / 190 (beam 4) of 256.
                  fp32 decode mode (decode_dtype = float32): 256 / 256 token-exact, every beam, scores within 1e-4
                  relative (measured 1.5e-6 / 3.2e-6).  bf16 product mode (round 6, after the residual sums of a decode
                  step stopped being rounded to bf16 in front of the LayerNorm: 209 / 203 of 256 token-exact against
                  the fp32 oracle, 205 / 198 against the bf16-storage oracle): NOT MORE misses than the bf16-storage
                  oracle has (+ 4), EVERY first divergence located and a near-tie of the fp32 oracle -- reproduced by the
                  bf16-storage oracle at that step, inside the oracle pair's own score distance at the last step the two
                  oracles share, or an fp32-oracle gap below BF16_TIE; no unexplained divergence is tolerated
                  (ACCEPTED_FAR lists exceptions by sentence id: none).  Per-divergence records go to
                  gpurun_out/fullsize_beam_*.json (copied to profiles/r06_parity_fullsize_beam_*.json).
"""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.fullsize import (fullsize_hp, fullsize_batch, fullsize_params, param_probe, SLICES,  # noqa: E402
                            beam_hp, beam_params, beam_sources)
from zero_amd.models._factory import get_core, reset_cores  # noqa: E402
from zero_amd.models import model as registry, load_all  # noqa: E402

load_all()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REPORT = os.path.join(os.path.dirname(GOLD), "..", "gpurun_out")


# element-wise tolerance of the three gradient slices (relative L2 against the bf16-storage oracle).  Slice 0 is a
# q/k block of the FIRST encoder layer's qkv_map: the end of the longest backward chain, and q/k gradients come out
# of the cancellation dS = P (dP - rowsum(dO o O)) whose rowsum the HIP kernels take from the bf16 O they stored
# (measured 3.4 % at d=512, 6.6 % at d=1024 -- the same against the fp32 oracle; every variable's gradient NORM is
# within 1 %, i.e. the difference is direction noise, not a missing term).  Slices 3 / 4 (first decoder layer's
# ffn enlarge, fourth encoder layer's o_map) sit 17 / 26 sub-layers deep in the backward chain: measured 3.5 % / 2.4 %
# with the 6-layer encoder, 7.2 % / 3.3 % behind the 12-layer encoder (its output already differs by bf16 noise).
# Slices 1 / 2 (last decoder layer, target embedding) are at the start of it: 0.5 % / 0.1 %.
# round 6: two runs of the bf16-storage ORACLE that differ only in the order of their fp32 sums are 3.2 / 0.45 / 0.08 / 2.8 / 2.7 %
# apart on these slices at the base size and 3.3 / 0.4 / 0.07 / 5.2 / 2.6 % with the 12-layer encoder
# (scripts/bf16_grad_noise_floor.py, profiles/r06_bf16_grad_noise_floor.txt): the bound is ~1.5x the storage model's own spread.
SLICE_TOL = (8e-2, 2e-2, 2e-2, 8e-2, 8e-2)      # round 4: 1e-1 -> 8e-2 (measured 3.0-6.6 %, attributed to the bf16 shadow
                                                  # weights in profiles/r03_grad_noise_attribution.json)


def _report(name, obj):
    try:
        os.makedirs(REPORT, exist_ok=True)
        with open(os.path.join(REPORT, "fullsize_%s.json" % name), "w") as f:
            json.dump(obj, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _train_case(name, **kw):
    fx = np.load(os.path.join(GOLD, name + ".npz"))
    reset_cores()
    hp = fullsize_hp(**kw)
    model = hp.model_name
    Pn = fullsize_params(hp, model)
    assert np.allclose(param_probe(Pn), fx["param_probe"], rtol=1e-12, atol=0), "regenerated parameters differ"
    src, tgt = fullsize_batch()
    out = registry.get_model(model).train_fn({"source": src, "target": tgt}, hp, initializer=Pn)
    torch.cuda.synchronize()
    loss = float(out["loss"].cpu())
    ps = out["per_sample_loss"].cpu().numpy()
    G = out["store"].export("grad")
    names = [str(n) for n in fx["names"]]
    rep = {"loss_hip": loss, "loss_f32": float(fx["f32_loss"]), "loss_bf16model": float(fx["bf16_loss"])}
    rep["loss_rel"] = abs(loss - rep["loss_f32"]) / abs(rep["loss_f32"])
    rep["per_sample_rel_max"] = float(np.abs(ps - fx["f32_per_sample"]).max() / np.abs(fx["f32_per_sample"]).max())
    gn = np.array([float(np.linalg.norm(G[k].astype(np.float64))) for k in names])
    gnorm = float(np.sqrt((gn ** 2).sum()))
    rep["gnorm_hip"], rep["gnorm_f32"], rep["gnorm_bf16model"] = gnorm, float(fx["f32_gnorm"]), float(fx["bf16_gnorm"])
    ref = fx["bf16_grad_norms"]
    big = ref >= 2e-2 * ref.max()
    rel = np.abs(gn - ref) / np.maximum(ref, 1e-30)
    rep["var_norm_rel_max_large"] = float(rel[big].max())
    rep["var_norm_rel_max_small"] = float(rel[~big & (ref > 1e-4 * ref.max())].max()) if (~big).any() else 0.0
    rep["var_norm_worst"] = names[int(np.argmax(np.where(big, rel, 0)))]
    order = np.argsort(-np.where(ref > 1e-4 * ref.max(), rel, 0))[:6]
    rep["var_norm_worst6"] = [[names[int(i)], float(rel[i]), float(ref[i]), float(gn[i]), float(fx["f32_grad_norms"][i])]
                              for i in order]
    rep["var_norm_rel_vs_f32_max"] = float((np.abs(gn - fx["f32_grad_norms"]) / np.maximum(fx["f32_grad_norms"], 1e-30))[big].max())
    for i, (k, rs, cs) in enumerate(SLICES):
        if k not in G:
            continue
        got = G[k][rs[0]:rs[1], cs[0]:cs[1]].astype(np.float64)
        for tag in ("bf16", "f32"):
            r = fx["%s_slice%d" % (tag, i)].astype(np.float64)
            rep["slice%d_rel_vs_%s" % (i, tag)] = float(np.linalg.norm(got - r) / np.linalg.norm(r))
    print(json.dumps(rep, indent=1, sort_keys=True))
    _report(name, rep)
    assert rep["loss_rel"] < 1e-3, rep
    assert rep["per_sample_rel_max"] < 2e-3, rep
    assert abs(gnorm - rep["gnorm_bf16model"]) / rep["gnorm_bf16model"] < 1e-2, rep
    assert abs(gnorm - rep["gnorm_f32"]) / rep["gnorm_f32"] < 3e-2, rep
    assert rep["var_norm_rel_max_large"] < 1.5e-2, rep       # round 4: 2e-2 -> 1.5e-2 (measured <= 1.1 %)
    assert rep["var_norm_rel_max_small"] < 1.5e-2, rep
    for i, tol in enumerate(SLICE_TOL):
        if "slice%d_rel_vs_bf16" % i in rep:
            assert rep["slice%d_rel_vs_bf16" % i] < tol, rep
    return rep


def test_base_config_loss_and_gradients():
    """BASELINE configs[1]: Transformer-base, B=64 x (64+64) tokens, V=32000."""
    _train_case("base_synth_seed1234")


def test_twelve_layer_encoder_config():
    """BASELINE configs[4] (the part with reference code): num_encoder_layer=12 (transformer.py:35-69)."""
    _train_case("enc12_synth_seed1234", num_encoder_layer=12)


def test_big_widths_six_layers():
    """BASELINE configs[2] widths: d=1024, F=4096, 16 heads, 6+6 layers, on one GPU."""
    _train_case("big_synth_seed1234", hidden_size=1024, embed_size=1024, filter_size=4096, num_heads=16)


NEAR_TIE_REL = 4e-3      # sanity cap on any accepted divergence (fraction of |score|); the criterion proper is the oracle pair, below


def _first_divergence(hip_trace, ref_scores, ref_idx, s_local, s_global, K):
    """First decode step at which the HIP search's 2K candidates of sentence s differ from the oracle's, and the gap
    the ORACLE saw there between the candidate it kept at that rank and the one the HIP path put there.  Up to that
    step both searches hold the same alive set, so flat indices (beam * V + token) are comparable."""
    T = min(len(hip_trace), ref_idx.shape[0])
    for t in range(T):
        r_idx = ref_idx[t, s_global]
        if r_idx[0] < 0:
            break
        h_idx = hip_trace[t][1][s_local]
        r_sc = ref_scores[t, s_global]
        # candidates parked at -inf / f32.min (finished or masked) carry no order information
        live = int(np.sum(r_sc[:2 * K] > -1e30))
        for j in range(live):
            if int(h_idx[j]) == int(r_idx[j]):
                continue
            where = np.nonzero(r_idx == h_idx[j])[0]
            gap = float(r_sc[j] - r_sc[int(where[0])]) if len(where) else None
            # a candidate below the oracle's last kept runner-up: the gap is AT LEAST the distance to that runner-up
            kept = r_sc[np.isfinite(r_sc) & (r_sc > -1e30)]
            lower = float(r_sc[j] - kept.min()) if not len(where) and kept.size else None
            return {"step": t, "rank": j, "oracle_flat_index": int(r_idx[j]), "hip_flat_index": int(h_idx[j]),
                    "oracle_score": float(r_sc[j]), "oracle_gap": gap, "oracle_gap_lower_bound": lower,
                    "oracle_pos_of_hip_candidate": int(where[0]) if len(where) else None,
                    "hip_gap_seen": float(hip_trace[t][0][s_local][j] - hip_trace[t][0][s_local][min(j + 1, 2 * K - 1)]),
                    "tolerance": NEAR_TIE_REL * max(abs(float(r_sc[j])), 1.0)}
    return None


def _pair_score_distance(f_sc, f_ix, b_sc, b_ix, t):
    """largest distance between the two oracles' scores of the candidates BOTH list at step t (comparable while the two
    hold the same alive set, i.e. up to and including the first step at which their kept candidates differ)"""
    moved = 0.0
    for a, ia in enumerate(f_ix[t]):
        w = np.nonzero(b_ix[t] == ia)[0]
        if ia >= 0 and len(w) and f_sc[t, a] > -1e30 and b_sc[t, int(w[0])] > -1e30:
            moved = max(moved, abs(float(f_sc[t, a]) - float(b_sc[t, int(w[0])])))
    return moved


def _bf16_oracle_view(fx, K, d, s_global):
    """What the oracle under the bf16 STORAGE model did at the step where the HIP search left the fp32 oracle's path
    (fixture keys bf16_trace_*).  Two facts, per divergence:
      bf16_oracle_same_choice          the bf16-storage oracle -- still on the common path up to that step -- has the
                                       HIP path's candidate at that rank: the miss is REPRODUCED by the checker itself;
      inside_oracle_pair_disagreement  the fp32 oracle's gap between the two candidates is no larger than the largest
                                       distance between the two oracles' scores of the SAME candidates (x 2: either
                                       candidate may move), measured at the divergence step when the bf16-storage oracle is
                                       still on the common path there, otherwise (round 6: this case used to be accepted
                                       unconditionally) at the LAST step the two oracles share on this sentence -- the step
                                       at which they part, whose tables are still over the same alive set."""
    gap = d.get("oracle_gap")
    if gap is None:
        gap = d.get("oracle_gap_lower_bound")
    if d.get("step") is None or gap is None:
        return {}
    t, j = d["step"], d["rank"]
    f_sc, f_ix = fx["trace_scores_k%d" % K][:, s_global], fx["trace_idx_k%d" % K][:, s_global]
    b_sc, b_ix = fx["bf16_trace_scores_k%d" % K][:, s_global], fx["bf16_trace_idx_k%d" % K][:, s_global]
    out = {}
    if t >= b_ix.shape[0] or b_ix[t, 0] < 0:
        return {"bf16_oracle_on_common_path": False}
    # the bf16-storage oracle is on the common path while its KEPT candidates (the first 2K) of all earlier steps
    # equal the fp32 oracle's
    first_off = next((u for u in range(t) if not np.array_equal(f_ix[u, :2 * K], b_ix[u, :2 * K])), None)
    common = first_off is None
    out["bf16_oracle_on_common_path"] = bool(common)
    if not common:
        # the two oracles part on this very sentence BEFORE the HIP search does (which followed the fp32 oracle longer
        # than the bf16-storage oracle did).  Their tables at the divergence step are over different alive sets; the
        # yardstick is their score distance at the parting step, the last one they share
        out["oracle_pair_parted_at_step"] = int(first_off)
        moved = _pair_score_distance(f_sc, f_ix, b_sc, b_ix, first_off)
        out["oracle_pair_score_distance_at_parting_step"] = moved
        out["inside_oracle_pair_disagreement"] = bool(d.get("oracle_gap") is not None and gap <= 2.0 * moved)
        return out
    out["bf16_oracle_flat_index"] = int(b_ix[t, j])
    out["bf16_oracle_same_choice"] = bool(int(b_ix[t, j]) == d["hip_flat_index"])
    moved = _pair_score_distance(f_sc, f_ix, b_sc, b_ix, t)
    out["oracle_pair_score_distance_at_step"] = moved
    out["inside_oracle_pair_disagreement"] = bool(d.get("oracle_gap") is not None and gap <= 2.0 * moved)
    return out


@pytest.mark.parametrize("K", [1, 4])
def test_aan_beam_search_base_size(K):
    """BASELINE configs[3] subset: transformer_aan, d=512, V=32000, 256 length-sorted sentences, eval batch 32, the bf16
    PRODUCT decode mode (the fp32 mode, which meets "token-id exact" outright, is the next test).
    Criterion (round 6): (a) not more hypotheses off the fp32 oracle than the bf16-storage oracle has off it (+ 4);
    (b) every hypothesis that is NOT token-exact is located -- the first step at which the HIP search's candidate table
    leaves the oracle's (fixture `trace_*`: the oracle's 2K kept candidates + eight runner-ups of every step) -- and is a
    near-tie of the fp32 oracle there: the bf16-storage oracle makes the same choice, or the fp32 oracle's gap between the
    two candidates is inside the oracle pair's own score distance, or it is below BF16_TIE (absolute, score units).  A
    divergence that is none of these fails the test -- that would be a bug in zk_dec_* / zk_beam_*.
    How much agreement with the bf16-storage oracle is available at all: two runs of THAT oracle that differ only in the order
    of their fp32 sums (float64-accumulated products, every bf16 rounding point unchanged) agree on 214 / 198 of the 256
    sentences (beam 1 / 4; scripts/bf16_oracle_noise_floor.py, profiles/r06_bf16_oracle_noise_floor.txt); the HIP path's
    205 / 198 are reported, not asserted."""
    from zero_amd.main import tower_infer_graph
    from zero_amd.search import decode_hypothesis
    fx = np.load(os.path.join(GOLD, "aan_base_beam.npz"))
    reset_cores()
    hp = beam_hp()
    hp.beam_size = K
    model = hp.model_name
    Pn = beam_params(hp, model)          # round 5: the weight set that decodes like a model (tests/fullsize.py)
    assert np.allclose(param_probe(Pn), fx["param_probe"], rtol=1e-12, atol=0)
    src = beam_sources()
    assert np.array_equal(src, fx["source"])
    get_core(hp, model, Pn)
    ref_seq, ref_score = fx["seqs_k%d" % K], fx["scores_k%d" % K]
    ref_tsc, ref_tix = fx["trace_scores_k%d" % K], fx["trace_idx_k%d" % K]
    exact = first8 = n = 0
    all_hyp = []
    prefix = []
    dscore = dscore_same = 0.0
    divergences = []
    for i in range(0, src.shape[0], 32):
        seqs, scores = tower_infer_graph({"source": src[i:i + 32]}, registry.get_model(model), hp)
        hyp = decode_hypothesis(seqs, hp)
        all_hyp.extend(hyp)
        ref_hyp = decode_hypothesis(ref_seq[i:i + 32], hp)
        miss = []
        for j, (a, b) in enumerate(zip(hyp, ref_hyp)):
            n += 1
            exact += int(list(a) == list(b))
            if list(a) == list(b):
                dscore_same = max(dscore_same, abs(float(scores[j, 0]) - float(ref_score[i + j, 0])))
            else:
                miss.append(j)
            m = 0
            while m < min(len(a), len(b)) and a[m] == b[m]:
                m += 1
            prefix.append(m / float(max(len(b), 1)))
            first8 += int(m >= min(8, len(b)))
        dscore = max(dscore, float(np.abs(scores[:, 0] - ref_score[i:i + 32, 0]).max()))
        if miss:
            # the same batch again with the per-step candidate tables recorded (host bookkeeping path: bit-identical
            # hypotheses, tests/test_gpu_model.py::test_device_resident_search_equals_host_bookkeeping)
            hp.search_trace = []
            seqs2, _ = tower_infer_graph({"source": src[i:i + 32]}, registry.get_model(model), hp)
            trace, hp.search_trace = hp.search_trace, None
            assert np.array_equal(np.asarray(seqs2), np.asarray(seqs)), "traced search differs from the default one"
            for j in miss:
                d = _first_divergence(trace, ref_tsc, ref_tix, j, i + j, K)
                d = dict(d or {"step": None, "oracle_gap": None}, sentence=i + j)
                d.update(_bf16_oracle_view(fx, K, d, i + j))
                if d.get("oracle_gap") is not None and d.get("rank") == 0:
                    # where this gap stands among the best-vs-second gaps of ALL (sentence, step) pairs of the fixture
                    allgaps = (ref_tsc[:, :, 0] - ref_tsc[:, :, 1])[ref_tix[:, :, 0] >= 0]
                    d["gap_rank_among_all_top2_gaps"] = int(np.sum(allgaps < d["oracle_gap"]))
                    d["all_top2_gaps"] = int(allgaps.size)
                divergences.append(d)
    # how common near-ties are in this random model: (sentence, step) pairs whose best two candidates are closer than
    # the tolerance, out of all pairs
    valid = ref_tix[:, :, 0] >= 0
    with np.errstate(invalid="ignore"):          # steps past a batch's end are NaN in the fixture
        top_gap = ref_tsc[:, :, 0] - ref_tsc[:, :, 1]
        tol = NEAR_TIE_REL * np.maximum(np.abs(ref_tsc[:, :, 0]), 1.0)
        near = int(np.sum(valid & (top_gap < tol)))
    rep = {"beam": K, "sentences": n, "token_exact": exact, "token_exact_rate": exact / float(n),
           "first8_rate": first8 / float(n), "mean_common_prefix_frac": float(np.mean(prefix)),
           "best_score_abs_diff_max": dscore, "best_score_abs_diff_max_same_hypothesis": dscore_same,
           "divergences": divergences, "near_tie_rel_tolerance": NEAR_TIE_REL,
           "oracle_top2_near_ties": near, "oracle_sentence_steps": int(valid.sum())}
    print(json.dumps(rep, sort_keys=True))
    _report("beam_k%d" % K, rep)
    # how often the two oracles (fp32 / bf16 storage) disagree with EACH OTHER on whole hypotheses: the yardstick for
    # "token-id exact" between two correct implementations that round at different points
    o_hyp = decode_hypothesis(ref_seq, hp)
    b_hyp = decode_hypothesis(fx["bf16_seqs_k%d" % K], hp)
    rep["oracle_fp32_vs_bf16storage_token_exact"] = int(sum(list(a) == list(b) for a, b in zip(o_hyp, b_hyp)))
    rep["token_exact_vs_bf16storage_oracle"] = int(sum(list(a) == list(b) for a, b in zip(all_hyp, b_hyp)))
    rep["reproduced_by_bf16_oracle"] = int(sum(bool(d.get("bf16_oracle_same_choice")) for d in divergences))
    rep["bracketed_by_oracle_pair"] = int(sum(bool(d.get("inside_oracle_pair_disagreement")) for d in divergences))
    print(json.dumps({k: rep[k] for k in ("oracle_fp32_vs_bf16storage_token_exact", "reproduced_by_bf16_oracle",
                                          "bracketed_by_oracle_pair")}))
    _report("beam_k%d" % K, rep)
    # On the fixture that decodes like a model the two ORACLES (fp32 / bf16 storage model) agree with each other on 206
    # (beam 1) and 190 (beam 4) of 256 hypotheses, their first differing tokens spread over positions 0 .. 43: "token-id
    # exact" between an fp32 and a bf16-storage implementation is an 80 % / 74 % property on a decode workload whose steps
    # matter.  The bf16 product path is held to the checker's own yardstick; exactness is the fp32 mode's job
    # (test_aan_beam_search_base_size_fp32_is_token_exact).
    n_pair = rep["oracle_fp32_vs_bf16storage_token_exact"]
    rep["criterion"] = {"misses_hip_vs_fp32": n - exact, "misses_oracle_pair": n - n_pair, "bf16_tie": BF16_TIE}
    _report("beam_k%d" % K, rep)
    # (a) the HIP path must not part from the fp32 oracle more often than the bf16-storage oracle does (+ 4: which of two
    #     near-tied candidates a rounding picks is a coin flip).  Measured (round 6, MI355X): beam 1: 47 misses against the
    #     oracle pair's 50; beam 4: 53 against 66 (round 5, with the residual sums rounded to bf16: 76 / 74).
    assert n - exact <= (n - n_pair) + 4, rep
    # ... and agrees with the bf16-storage oracle at least as often as the fp32 oracle does, less the same allowance
    # (measured 205 / 198 of 256)
    assert rep["token_exact_vs_bf16storage_oracle"] >= n_pair - 12, rep
    assert dscore_same < 0.3, rep
    # (b) every divergence is located (a step, a rank) ...
    for d in divergences:
        assert d["step"] is not None, ("unlocated divergence", d)
    # ... and is a near-tie for a bf16 implementation: reproduced by the bf16-storage oracle, inside the oracle pair's own
    # score distance, or closer than BF16_TIE (absolute, in score units: the logits of this weight set carry ~1e-2 of
    # bf16 noise -- softmax rows of norm 3, the EOS row of norm 6 -- against scores of -3 .. -30).  A HIP candidate below
    # the oracle's eight runner-ups has only a LOWER bound of its gap on record: it counts as far unless the bf16-storage
    # oracle reproduces it.
    def explained(d):
        if d.get("bf16_oracle_same_choice"):
            return True
        if d.get("oracle_gap") is None:
            return False
        return bool(d.get("inside_oracle_pair_disagreement")) or d["oracle_gap"] < BF16_TIE
    far = [d for d in divergences if not explained(d)]
    rep["criterion"]["far_divergences"] = far
    rep["criterion"]["largest_oracle_gap"] = max([d["oracle_gap"] for d in divergences if d.get("oracle_gap") is not None] or [0.0])
    _report("beam_k%d" % K, rep)
    assert [d["sentence"] for d in far if d["sentence"] not in ACCEPTED_FAR[K]] == [], far
    # the divergences are not all at step 0
    if len(divergences) >= 8:
        assert sum(1 for d in divergences if d["step"] > 0) >= len(divergences) // 3, [d["step"] for d in divergences]


BF16_TIE = 0.08          # (measured, round 6: the largest fp32-oracle gap at a first divergence is 0.046 at beam 1, 0.055 at beam 4;
                         #  round 5, with the extra rounding of the residual sums: 0.059 / 0.106 under a bound of 0.15)
ACCEPTED_FAR = {1: (), 4: ()}      # sentence ids of divergences accepted although unexplained: none


def _hyp_stats(seqs, src):
    """share of best hypotheses that end in an EOS / of positions that repeat the previous token (the fixture's own
    `stats_k*` hold the oracle's figures; make_fullsize_golden.py beam_stats)"""
    ends = reps = tot = 0
    for s_ in np.asarray(seqs)[:, 0]:
        s_ = [int(x) for x in s_]
        L = s_.index(2) + 1 if 2 in s_ else len([x for x in s_ if x != 0])
        ends += int(2 in s_)
        reps += sum(1 for a, b in zip(s_[1:L], s_[:L - 1]) if a == b)
        tot += max(L - 1, 1)
    return ends / float(len(seqs)), reps / float(tot)


@pytest.mark.parametrize("K", [1, 4])
def test_aan_beam_search_base_size_fp32_is_token_exact(K):
    """north_star: "token-id exact for greedy decode" against the fp32 reference.  In the fp32 decode mode
    (decode_dtype = float32: fp32 masters / activations / accumulation, zk_f32_*) EVERY one of the 256 best hypotheses
    must equal the fp32 oracle's token for token, beam 1 and beam 4, scores within 1e-4 relative -- on a fixture whose
    hypotheses end in an EOS at lengths around the source length (EOS routing, finished-set merging and the stop bound of
    search.py:85-113, 192-228 are exercised at d = 512, V = 32000)."""
    from zero_amd.main import tower_infer_graph
    from zero_amd.search import decode_hypothesis
    fx = np.load(os.path.join(GOLD, "aan_base_beam.npz"))
    reset_cores()
    hp = beam_hp()
    hp.beam_size = K
    hp.decode_dtype = "float32"
    model = hp.model_name
    Pn = beam_params(hp, model)
    assert np.allclose(param_probe(Pn), fx["param_probe"], rtol=1e-12, atol=0)
    src = beam_sources()
    get_core(hp, model, Pn)
    ref_seq, ref_score = fx["seqs_k%d" % K], fx["scores_k%d" % K]
    ref_tsc, ref_tix = fx["trace_scores_k%d" % K], fx["trace_idx_k%d" % K]
    exact = allbeams = n = 0
    rel = 0.0
    misses = []
    steps = 0
    import time as _time
    t0 = _time.time()
    for i in range(0, src.shape[0], 32):
        seqs, scores = tower_infer_graph({"source": src[i:i + 32]}, registry.get_model(model), hp)
        seqs = np.asarray(seqs)
        steps += seqs.shape[2]
        hyp = decode_hypothesis(seqs, hp)
        ref_hyp = decode_hypothesis(ref_seq[i:i + 32], hp)
        L = min(seqs.shape[2], ref_seq.shape[2])
        bad = []
        for j, (a, b) in enumerate(zip(hyp, ref_hyp)):
            n += 1
            same = list(a) == list(b)
            exact += int(same)
            allbeams += int(np.array_equal(seqs[j, :, :L], ref_seq[i + j, :, :L]))
            if same:
                rel = max(rel, abs(float(scores[j, 0]) - float(ref_score[i + j, 0])) / max(abs(float(ref_score[i + j, 0])), 1e-6))
            else:
                bad.append(j)
        if bad:
            hp.search_trace = []
            tower_infer_graph({"source": src[i:i + 32]}, registry.get_model(model), hp)
            trace, hp.search_trace = hp.search_trace, None
            for j in bad:
                d = _first_divergence(trace, ref_tsc, ref_tix, j, i + j, K)
                misses.append(dict(d or {"step": None}, sentence=i + j))
    wall = _time.time() - t0
    ends, reps = _hyp_stats(ref_seq, src)
    rep = {"beam": K, "sentences": n, "token_exact": exact, "all_beams_token_exact": allbeams,
           "best_score_rel_diff_max": rel, "misses": misses, "oracle_eos_terminated_frac": ends, "oracle_repeat_frac": reps,
           "decode_wall_s_incl_startup": wall, "decode_steps_sum_over_batches": steps}
    print(json.dumps(rep, sort_keys=True))
    _report("beam_fp32_k%d" % K, rep)
    assert exact == n, rep
    assert rel < 1e-4, rep
