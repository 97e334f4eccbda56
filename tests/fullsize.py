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


def fullsize_params(hp, model, seed=1234):
    from oracle import ref_torch as rt
    return perturb(rt.init_params(hp, model, seed=seed), np.random.default_rng(seed + 1))


def param_probe(Pn):
    """A few numbers that pin the regenerated parameters (sum and sum of squares of five variables)."""
    keys = sorted(Pn.keys())
    pick = [keys[0], keys[len(keys) // 3], keys[len(keys) // 2], keys[-2], "tgt_embedding"]
    return np.array([[float(np.asarray(Pn[k], np.float64).sum()), float((np.asarray(Pn[k], np.float64) ** 2).sum())]
                     for k in pick])


def beam_hp():
    hp = fullsize_hp("transformer_aan")
    hp.override_from_dict(dict(beam_size=4, decode_alpha=0.6, decode_length=50, search_mode="cache",
                               eval_batch_size=32))
    return hp


BEAM_SENTENCES = 256      # full-size beam fixture (round 4: 256 sentences = 8 eval batches; 64 before)


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
