# coding: utf-8
"""Builders shared by tests/golden/make_fullsize_golden.py (build container, runs the oracle) and
tests/test_gpu_fullsize.py (GPU box, runs the HIP path): the BASELINE-size configurations, the synthetic
batch of SURVEY.md 8(d) and the deterministic parameters both sides regenerate instead of storing."""
import numpy as np

from zero_amd.config import transformer_base_params, SyntheticVocab
from tests.common import perturb

B, LS, LT, V = 64, 64, 64, 32000

# (variable, row range, column range) of the gradient slices kept in the fixtures
SLICES = [
    ("encoder/layer_0/self_attention/dot_attention/qkv_map/W_0_0", (0, 48), (500, 548)),
    ("decoder/layer_5/feed_forward/ffn_layer/output/W_0_0", (1000, 1048), (100, 148)),
    ("tgt_embedding", (0, 64), (0, 64)),
    ("decoder/layer_0/feed_forward/ffn_layer/enlarge/W_0_0", (100, 148), (700, 748)),
    ("encoder/layer_3/self_attention/dot_attention/o_map/W_0_0", (0, 48), (0, 48)),
]


def fullsize_hp(model="transformer", **kw):
    hp = transformer_base_params(dropout=0.0, relu_dropout=0.0, residual_dropout=0.0, attention_dropout=0.0,
                                 update_cycle=1, token_size=4096, model_name=model, scope_name="fs_" + model)
    hp.override_from_dict(kw)
    hp.src_vocab = SyntheticVocab(V)
    hp.tgt_vocab = SyntheticVocab(V)
    return hp


def fullsize_batch(seed=1234):
    """ids ~ U{3..V-1}, last column eos(2), no padding (the bench batch of rank 0)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(3, V, size=(B, LS), dtype=np.int64)
    tgt = rng.integers(3, V, size=(B, LT), dtype=np.int64)
    src[:, -1] = 2
    tgt[:, -1] = 2
    return src, tgt


def fullsize_params(hp, model, seed=1234, init=None):
    """init: (hp, model, seed) -> {name: array}; default the oracle's init_params.  zero_amd.variables.initial_values
    draws the same values from the same numpy stream (tests/test_oracle.py holds the two equal), which lets bench.py build
    the decode fixture's weight set without touching oracle/."""
    if init is None:
        from oracle import ref_torch as rt
        init = lambda hp_, model_, seed_: rt.init_params(hp_, model_, seed=seed_)
    return perturb(init(hp, model, seed), np.random.default_rng(seed + 1))


def param_probe(Pn):
    """A few numbers that pin the regenerated parameters (sum and sum of squares of five variables)."""
    keys = sorted(Pn.keys())
    pick = [keys[0], keys[len(keys) // 3], keys[len(keys) // 2], keys[-2], "tgt_embedding"]
    return np.array([[float(np.asarray(Pn[k], np.float64).sum()), float((np.asarray(Pn[k], np.float64) ** 2).sum())]
                     for k in pick])


def beam_hp():
    """BASELINE configs[3] decode workload.  Round 5: the softmax embedding is NOT tied to the target embedding and the
    scope initialiser runs at gain 0.1 -- see beam_params() for why (a gain-1 random post-LN stack is a constant function
    of its prefix)."""
    hp = fullsize_hp("transformer_aan")
    hp.override_from_dict(dict(beam_size=4, decode_alpha=0.6, decode_length=50, search_mode="cache",
                               eval_batch_size=32, shared_target_softmax_embedding=False, initializer_gain=0.1))
    return hp


BEAM_SENTENCES = 256      # full-size beam fixture (round 4: 256 sentences = 8 eval batches; 64 before)

# ---- the decode fixture's weight set: a random Transformer that DECODES LIKE A MODEL -------------
# Measured on the oracle: with the recipe's initialiser (gain 1) the decoder feature of a random 6-layer post-LN stack
# moves by 5 % of its norm when the previous token changes (every ReLU feed-forward adds a constant vector that the next
# LayerNorm trades against the token's share), so greedy search emits one token until the length cap whatever the softmax
# table is -- the round-4 fixture (0 of 256 hypotheses with an EOS, 99 % repeats, every divergence at step 0).  This
# weight set keeps the architecture, sizes and every code path and makes the function of the prefix non-trivial:
#   * scope initialiser at gain 0.1 (hp.initializer_gain; the sub-layer outputs no longer drown the residual stream:
#     the feature moves by 45 % with the previous token), target embedding rows x BEAM_EMB_SCALE, cross-attention
#     projections x BEAM_CROSS_SCALE (source dependence: sentences decode differently);
#   * softmax_embedding (untied): row w = BEAM_SHARPEN x the sum of the (unscaled) target embeddings of w's
#     BEAM_SUCCESSORS predecessors under random successor maps -- a sparse bigram model read off the residual stream, with
#     the average-attention sub-layer mixing in the successors of earlier tokens: top-1 probabilities 0.05 .. 0.8;
#   * the EOS row is FITTED (tests/golden/make_beam_eos_row.py: ridge regression on oracle features of 96 other
#     sentences) so that logit(eos) ~ top logit + BEAM_EOS_KAPPA x (t - (source length - 1)): the encoder's mean timing
#     signal carries the source length into every decoder position through the (near-uniform) cross attention, the
#     decoder's own timing signal carries t.  Stored as data: tests/golden/aan_base_beam_eos_row.npy (512 floats).
# Result (fixture header, make_fullsize_golden.py beam): hypotheses end in EOS at lengths spread around the source length.
BEAM_EMB_SCALE = 2.0
BEAM_CROSS_SCALE = 2.0
BEAM_SHARPEN = 1.5
BEAM_SUCCESSORS = 4
BEAM_EOS_KAPPA = 0.4
BEAM_EOS_RIDGE = 300.0
BEAM_EOS_ROW = "aan_base_beam_eos_row.npy"


def beam_params(hp, model="transformer_aan", seed=1234, eos_row="file", init=None):
    """Parameters of the decode fixture (see above), regenerated on both sides from numpy streams + the stored EOS row.
    eos_row: "file" (tests/golden/aan_base_beam_eos_row.npy), an array, or None (row of zeros: make_beam_eos_row.py)."""
    import os
    Pn = fullsize_params(hp, model, seed, init=init)
    rng = np.random.default_rng(seed + 3087)
    Et = Pn["tgt_embedding"]
    Es = np.zeros_like(Pn["softmax_embedding"])
    for _ in range(BEAM_SUCCESSORS):
        succ = 3 + rng.permutation(V - 3)              # successor map on the ordinary ids (pad / unk / eos have none)
        Es[succ] += Et[3:V]
    Es *= np.float32(BEAM_SHARPEN)
    if isinstance(eos_row, str):
        eos_row = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", BEAM_EOS_ROW))
    if eos_row is not None:
        Es[2] = np.asarray(eos_row, np.float32)
    Pn["softmax_embedding"] = Es
    Pn["tgt_embedding"] = (Et * np.float32(BEAM_EMB_SCALE)).astype(np.float32)
    for k in Pn:
        if "/cross_attention/" in k and k.endswith("W_0_0"):
            Pn[k] = (Pn[k] * np.float32(BEAM_CROSS_SCALE)).astype(np.float32)
    return Pn


def beam_sources(n=BEAM_SENTENCES, seed=1234):
    """SURVEY.md 8(d) decode input: lengths ~ clipped Normal(28, 14) in [4, 100] + eos, length-sorted, padded."""
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(28, 14, n)), 4, 100).astype(int)
    lens = np.sort(lens)
    src = np.zeros((n, int(lens.max()) + 1), dtype=np.int64)
    for i, l in enumerate(lens):
        src[i, :l] = rng.integers(3, V, l)
        src[i, l] = 2
    return src
