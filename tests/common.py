"""Shared builders for the model-level tests (CPU and GPU)."""
import numpy as np

from zero_amd.config import default_params, SyntheticVocab


def make_hp(model, H=128, F=256, heads=2, layers=2, Vs=120, Vt=104, **kw):
    hp = default_params()
    hp.override_from_dict(dict(
        hidden_size=H, embed_size=H, filter_size=F, num_heads=heads, num_encoder_layer=layers,
        num_decoder_layer=layers, dropout=0., relu_dropout=0., residual_dropout=0., attention_dropout=0.,
        label_smooth=0.1, model_name=model, scope_name="t_" + model, max_relative_position=4,
        initializer="uniform_unit_scaling", initializer_gain=1.0, beam_size=4, decode_length=6,
        lrate_strategy="noam", lrate=1.0, warmup_steps=4000, beta1=0.9, beta2=0.98, epsilon=1e-8,
        clip_grad_norm=0.0))
    hp.override_from_dict(kw)
    hp.src_vocab = SyntheticVocab(Vs)
    hp.tgt_vocab = SyntheticVocab(Vt)
    return hp


def make_batch(rng, B, Ls, Lt, Vs, Vt, full_first=True):
    """Ragged batch: every row ends with eos(2) then pad(0); row 0 has full length."""
    src = np.zeros((B, Ls), dtype=np.int64)
    tgt = np.zeros((B, Lt), dtype=np.int64)
    for b in range(B):
        ls = int(rng.integers(2, Ls + 1))
        lt = int(rng.integers(2, Lt + 1))
        if b == 0 and full_first:
            ls, lt = Ls, Lt
        src[b, :ls - 1] = rng.integers(3, Vs, ls - 1)
        src[b, ls - 1] = 2
        tgt[b, :lt - 1] = rng.integers(3, Vt, lt - 1)
        tgt[b, lt - 1] = 2
    return src, tgt


def perturb(Pn, rng):
    """Make biases / LN parameters non-trivial so that every term is exercised."""
    for k in Pn:
        if k.endswith("b_0") or k.endswith("offset"):
            Pn[k] = rng.normal(0, 0.1, Pn[k].shape).astype(Pn[k].dtype)
        if k.endswith("scale"):
            Pn[k] = (1 + rng.normal(0, 0.1, Pn[k].shape)).astype(Pn[k].dtype)
    return Pn
