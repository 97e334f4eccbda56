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


def host_sort_arrays(ids, shift):
    """The checker of zk_batch_prep's grouping (numpy, stable): token rows grouped by embedding id.  shift: row (b, t)
    carries id[b, t-1]; rows with t == 0 have no embedding.  Returns (rows_sorted, seg, uid)."""
    ids = np.asarray(ids)
    B, L = ids.shape
    flat = ids.reshape(-1)
    rows = np.arange(B * L, dtype=np.int64)
    if shift:
        rows = rows[rows % L != 0]
        tok = flat[rows - 1]
    else:
        tok = flat
    order = np.argsort(tok, kind="stable")
    rows_sorted, tok_sorted = rows[order], tok[order]
    uid, first = np.unique(tok_sorted, return_index=True)
    seg = np.concatenate([first, [len(tok_sorted)]])
    return rows_sorted.astype(np.int32), seg.astype(np.int32), uid.astype(np.int32)


def device_sort_arrays(eng, name, ids, shift, max_id=0):
    """zk_batch_prep on one side (GPU tests): the dict zk_embed_bwd_sorted takes.  A shifted side goes in as the target
    of a one-column dummy source."""
    import torch
    from zero_amd.models._core import TransformerCore
    import types
    ids = np.asarray(ids)
    B, L = ids.shape
    fake = types.SimpleNamespace(eng=eng)
    dev = eng.buf("t.ids." + name, (B, L), torch.int32)
    dev.copy_(torch.from_numpy(ids.astype(np.int32)))
    srt = TransformerCore._sort_buffers(fake, name, B * L)
    if shift:
        dummy = eng.buf("t.ids.dummy", (B, 1), torch.int32)
        dummy.fill_(1)
        batch = {"B": B, "Ls": 1, "Lt": L, "src": dummy, "tgt": dev, "tgt_sort": srt, "max_id": max_id}
    else:
        batch = {"B": B, "Ls": L, "src": dev, "src_sort": srt, "max_id": max_id}
    eng.batch_prep(batch)
    return srt
