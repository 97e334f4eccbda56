# coding: utf-8
"""ORACLE (test infrastructure, never shipped, never measured as the product).

A CPU restatement, in plain unfused torch ops (fp32 or fp64, autograd for the
backward), of the reference's Transformer hot path.  Every function cites the
reference file:line it follows.

PARITY UNPINNED: the reference is TensorFlow-1.x graph code; TensorFlow cannot
be imported in the build container and the reference ships no tests or golden
vectors, so TF1 itself was never executed.  This restatement is pinned instead
by (i) an independent numpy-fp64 restatement (oracle/ref_numpy.py) that must
agree to 1e-9, (ii) the reference's own dual code paths (cache vs dev search
mode, aan_mask True vs False, train-time vs incremental decoding), (iii)
analytic known-answer tests and (iv) finite-difference gradient checks -- see
tests/test_oracle_*.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Granularity is deliberately that of func.py (one torch op per TF op), so that
timing it on host cores is a fair "port" CPU baseline.
"""

import copy
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

F32_MIN = float(np.finfo(np.float32).min)


# --------------------------------------------------------------------------
# variables: names, shapes, initialisation (host side)
# --------------------------------------------------------------------------
def _attn_vars(prefix, H, self_att, rpr, nrel, d):
    """func.py:194-216,278 (+ modules/rpr.py:49-57) variable set of one
    dot_attention scope."""
    v = []
    p = prefix + "/dot_attention/"
    if self_att:
        v += [(p + "qkv_map/W_0_0", (H, 3 * H), "w"), (p + "qkv_map/b_0", (3 * H,), "zeros")]
    else:
        for m in ("q_map", "k_map", "v_map"):
            v += [(p + m + "/W_0_0", (H, H), "w"), (p + m + "/b_0", (H,), "zeros")]
    if rpr:
        v += [(p + "rpr_keys/embeddings", (nrel, d), "w")]
        v += [(p + "rpr_values/embeddings", (nrel, d), "w")]
    v += [(p + "o_map/W_0_0", (H, H), "w"), (p + "o_map/b_0", (H,), "zeros")]
    v += [(prefix + "/layer_norm/scale", (H,), "ones"), (prefix + "/layer_norm/offset", (H,), "zeros")]
    return v


def _ffn_vars(prefix, H, Fs):
    """func.py:327-338 under scope feed_forward (transformer.py:60-69)."""
    p = prefix + "/ffn_layer/"
    return [(p + "enlarge/W_0_0", (H, Fs), "w"), (p + "enlarge/b_0", (Fs,), "zeros"),
            (p + "output/W_0_0", (Fs, H), "w"), (p + "output/b_0", (H,), "zeros"),
            (prefix + "/layer_norm/scale", (H,), "ones"), (prefix + "/layer_norm/offset", (H,), "zeros")]


def variable_specs(hp, model_name):
    """Ordered (name, shape, kind, layer) list of the trainable variables the
    reference creates for ``model_name`` (names relative to the model scope;
    transformer.py:16-33,88-102,184-192; transformer_aan.py:165-192;
    transformer_rpr.py:54-55,144-146,167-169)."""
    H, E, Fs = hp.hidden_size, hp.embed_size, hp.filter_size
    assert H == E, "the reference adds embeddings to H-wide layers"
    d = H // hp.num_heads
    rpr = model_name == "transformer_rpr"
    aan = model_name == "transformer_aan"
    fuse = model_name == "transformer_fuse"
    nrel = 2 * hp.max_relative_position + 1
    Vs, Vt = hp.src_vocab.size(), hp.tgt_vocab.size()
    specs = []
    if hp.shared_source_target_embedding:
        specs.append(("embedding", (Vs, E), "embed", None))
    else:
        specs.append(("src_embedding", (Vs, E), "embed", None))
    specs.append(("bias", (E,), "w", None))
    for l in range(hp.num_encoder_layer):
        pre = "encoder/layer_%d" % l
        for n, s, k in _attn_vars(pre + "/self_attention", H, True, rpr, nrel, d):
            specs.append((n, s, k, l))
        for n, s, k in _ffn_vars(pre + "/feed_forward", H, Fs):
            specs.append((n, s, k, l))
    if not hp.shared_source_target_embedding:
        specs.append(("tgt_embedding", (Vt, E), "embed", None))
    for l in range(hp.num_decoder_layer):
        pre = "decoder/layer_%d" % l
        if fuse:
            # transformer_fuse.py:131-160: one merged attention sub-layer, then the FFN
            for n, s, k in _attn_vars(pre + "/fuse_attention", H, False, False, nrel, d):
                specs.append((n, s, k, l))
            for n, s, k in _ffn_vars(pre + "/feed_forward", H, Fs):
                specs.append((n, s, k, l))
            continue
        if aan:
            a = pre + "/average_attention"
            if hp.use_ffn:
                p = a + "/ffn_layer/"
                specs += [(p + "enlarge/W_0_0", (H, Fs), "w", l), (p + "enlarge/b_0", (Fs,), "zeros", l),
                          (p + "output/W_0_0", (Fs, H), "w", l), (p + "output/b_0", (H,), "zeros", l)]
            specs += [(a + "/z_project/W_0_0", (2 * H, 2 * H), "w", l),
                      (a + "/z_project/b_0", (2 * H,), "zeros", l),
                      (a + "/layer_norm/scale", (H,), "ones", l),
                      (a + "/layer_norm/offset", (H,), "zeros", l)]
        else:
            for n, s, k in _attn_vars(pre + "/self_attention", H, True, rpr, nrel, d):
                specs.append((n, s, k, l))
        for n, s, k in _attn_vars(pre + "/cross_attention", H, False, rpr, nrel, d):
            specs.append((n, s, k, l))
        for n, s, k in _ffn_vars(pre + "/feed_forward", H, Fs):
            specs.append((n, s, k, l))
    if not hp.shared_source_target_embedding and not hp.shared_target_softmax_embedding:
        specs.append(("softmax_embedding", (Vt, E), "embed", None))
    return specs


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    return shape[0], shape[1]


def _scope_init(rng, shape, kind, gain):
    """modules/initializer.py:11-32 (TF1 initializer semantics)."""
    fi, fo = _fans(shape)
    if kind == "uniform":
        return rng.uniform(-gain, gain, size=shape)
    if kind == "normal":
        return rng.normal(0.0, gain, size=shape)
    if kind == "uniform_unit_scaling":
        lim = math.sqrt(3.0 * gain / ((fi + fo) / 2.0))
        return rng.uniform(-lim, lim, size=shape)
    if kind == "normal_unit_scaling":
        std = math.sqrt(gain / ((fi + fo) / 2.0))
        return np.clip(rng.normal(0.0, std, size=shape), -2 * std, 2 * std)
    lim = math.sqrt(6.0 / (fi + fo))  # glorot_uniform fallback
    return rng.uniform(-lim, lim, size=shape)


def init_params(hp, model_name, seed=1234, dtype=np.float32):
    """Draw every variable from the reference's *distribution* (bitwise TF RNG
    parity is impossible).  transformer.py:18,90 (embeddings N(0,H^-0.5));
    main.py:26 (scope initializer); transformer.py:38-44 (deep init)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    H = hp.hidden_size
    for name, shape, kind, layer in variable_specs(hp, model_name):
        if kind == "embed":
            v = rng.normal(0.0, H ** -0.5, size=shape)
        elif kind == "zeros":
            v = np.zeros(shape)
        elif kind == "ones":
            v = np.ones(shape)
        else:
            if layer is not None and hp.deep_transformer_init:
                v = _scope_init(rng, shape, "uniform_unit_scaling",
                                hp.initializer_gain * (layer + 1) ** -0.5)
            else:
                v = _scope_init(rng, shape, hp.initializer, hp.initializer_gain)
        out[name] = np.asarray(v, dtype=dtype)
    return out


def to_torch(params_np, dtype=torch.float32, requires_grad=False):
    out = OrderedDict()
    for k, v in params_np.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


# --------------------------------------------------------------------------
# func.py restated
# --------------------------------------------------------------------------
TRACE_RUNNER_UPS = 8      # candidates kept beyond the 2K of search.py:172-176 in hp.search_trace (checker aid)


class Cfg(object):
    """dtype.py:12-15 constants + run-time switches of the restatement."""
    eps = 1e-8
    inf = 1e8
    # Storage model of the MI355X build (a property of the CHECKER, not of the reference): when set, every
    # tensor the HIP path keeps in HBM as bf16 -- matrix weights read by the GEMMs, the output of every
    # linear / attention / LayerNorm / embedding, the softmax probabilities that feed P.V, and the gradients
    # of those tensors in the backward -- is rounded to bf16 (round to nearest even) at the same point; all
    # arithmetic and every accumulation stay fp32.  Lets the GPU tests assert a tight per-variable gradient
    # tolerance instead of the 12 % the pure-fp32 comparison needs.
    store_bf16 = False
    # attribution aid (scripts/grad_noise_attribution.py): when not None, only the storage sites named here are rounded
    # (weights, linear, probs, scores, attn_out, ln, embed, aan, logits); None = every site
    store_sites = None


def _bf16(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _on(site):
    return Cfg.store_bf16 and (Cfg.store_sites is None or site in Cfg.store_sites)


def _st(x, site="linear"):
    """activation stored as bf16 (value and its gradient)."""
    return _RoundBoth.apply(x) if _on(site) else x


def _st_fwd(x, site="weights"):
    """value rounded (bf16 shadow weights, probabilities), gradient kept fp32."""
    return _RoundFwd.apply(x) if _on(site) else x


def _st_bwd(x, site="scores"):
    """value kept fp32 (scores / logits live in registers or fp32), gradient stored bf16."""
    return _RoundBwd.apply(x) if _on(site) else x


def dropout(x, p, training=True):
    """util.py:75-79 valid_apply_dropout (tf.nn.dropout keep_prob=1-p)."""
    if p is not None and 0.0 < p <= 1.0 and training:
        return F.dropout(x, p, True)
    return x


def linear(x, P, scope, bias=True):
    """func.py:14-65 (single input, single output, no ln)."""
    W = _st_fwd(P[scope + "/W_0_0"])
    shp = x.shape
    o = torch.matmul(x.reshape(-1, shp[-1]), W)
    if bias:
        o = o + P[scope + "/b_0"]
    return _st(o.reshape(*shp[:-1], W.shape[1]))


def split_heads(x, n):
    """func.py:68-85."""
    b, l, c = x.shape
    return x.reshape(b, l, n, c // n).permute(0, 2, 1, 3)


def combine_heads(x):
    """func.py:88-104."""
    b, h, l, c = x.shape
    return x.permute(0, 2, 1, 3).reshape(b, l, h * c)


def rel_pos_matrix(len_x, len_y, max_rel):
    """modules/rpr.py:62-75: clip(i - j) + max_rel."""
    d = torch.arange(len_x)[:, None] - torch.arange(len_y)[None, :]
    return torch.clamp(d, -max_rel, max_rel) + max_rel


def rel_pos_embeddings(P, name, len_x, len_y, max_rel, last=None):
    """modules/rpr.py:44-59."""
    m = rel_pos_matrix(len_x, len_y, max_rel)
    if last is not None:
        m = m[-last:]
    return P[name + "/embeddings"][m]


def relative_attention_inner(x, y, z, transpose):
    """modules/rpr.py:10-41."""
    if transpose:
        xy = torch.matmul(x, y.transpose(-1, -2))
        xz = torch.einsum("bhqd,qkd->bhqk", x, z)
    else:
        xy = torch.matmul(x, y)
        xz = torch.einsum("bhqk,qkd->bhqd", x, z)
    return xy + xz


def dot_attention(query, memory, mem_mask, H, P, scope, num_heads, cache=None,
                  drop=None, use_rpr=False, max_rel=16, decode_step=None,
                  training=True, fuse_mask=None):
    """func.py:164-286 (out_map=True).  fuse_mask (func.py:258-275): the [B,L,L] averaging matrix
    during training, the time step (number) while decoding."""
    if fuse_mask is not None:
        assert memory is not None, 'Fuse mechanism only applied with cross-attention'
    scope = scope + "/dot_attention"
    if memory is None:
        h = linear(query, P, scope + "/qkv_map")
        q, k, v = torch.split(h, H, dim=-1)
        if cache is not None:
            k = torch.cat([cache['k'], k], dim=1)
            v = torch.cat([cache['v'], v], dim=1)
            cache = {'k': k, 'v': v}
    else:
        q = linear(query, P, scope + "/q_map")
        if cache is not None and ('mk' in cache and 'mv' in cache):
            k, v = cache['mk'], cache['mv']
        else:
            k = linear(memory, P, scope + "/k_map")
            v = linear(memory, P, scope + "/v_map")
        if cache is not None:
            cache['mk'] = k
            cache['mv'] = v
    q = split_heads(q, num_heads)
    k = split_heads(k, num_heads)
    v = split_heads(v, num_heads)
    q = q * (H // num_heads) ** (-0.5)

    q_len = q.shape[2] if decode_step is None else decode_step + 1
    r_lst = None if decode_step is None else 1
    if use_rpr:
        r = rel_pos_embeddings(P, scope + "/rpr_keys", q_len, k.shape[2], max_rel, r_lst)
        logits = relative_attention_inner(q, k, r, True)
    else:
        logits = torch.matmul(q, k.transpose(-1, -2))
    if mem_mask is not None:
        logits = logits + mem_mask
    weights = torch.softmax(_st_bwd(logits), dim=-1)
    dweights = _st_fwd(dropout(weights, drop, training), "probs")
    if use_rpr:
        r = rel_pos_embeddings(P, scope + "/rpr_values", q_len, k.shape[2], max_rel, r_lst)
        o = relative_attention_inner(dweights, v, r, False)
    else:
        o = torch.matmul(dweights, v)
    o = _st(combine_heads(o), "attn_out")
    if fuse_mask is not None:
        v_q = linear(query, P, scope + "/v_map")        # shares v_map with the memory side
        if cache is not None and 'aan' in cache:
            aan_o = (v_q + cache['aan']) / float(fuse_mask + 1)
        else:
            aan_o = torch.matmul(fuse_mask, v_q)
        if cache is not None:
            cache['aan'] = v_q if 'aan' not in cache else v_q + cache['aan']
        o = _st(o + aan_o, "attn_out")
    o = linear(o, P, scope + "/o_map")
    return {'weights': weights, 'output': o, 'cache': cache}


def layer_norm(x, P, scope):
    """func.py:289-303 (biased variance, eps inside rsqrt)."""
    scale = P[scope + "/layer_norm/scale"]
    offset = P[scope + "/layer_norm/offset"]
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return _st(scale * (x - mean) * torch.rsqrt(var + Cfg.eps) + offset, "ln")


def residual_fn(x, y, drop=None, training=True):
    """func.py:321-324."""
    return x + dropout(y, drop, training)


def ffn_layer(x, P, scope, drop=None, training=True):
    """func.py:327-338 (ReLU)."""
    hidden = torch.relu(linear(x, P, scope + "/ffn_layer/enlarge"))
    hidden = dropout(hidden, drop, training)
    return linear(hidden, P, scope + "/ffn_layer/output")


def timing_signal(length, channels, dtype, time=None):
    """func.py:341-369: concat([sin, cos]) with (channels/2 - 1) denominator."""
    if time is None:
        position = torch.arange(length, dtype=dtype)
    else:
        position = torch.tensor([float(time)], dtype=dtype)
    nts = channels // 2
    inc = math.log(1.0e4 / 1.0) / (float(nts) - 1)
    inv = torch.exp(torch.arange(nts, dtype=dtype) * -inc)
    st = position[:, None] * inv[None, :]
    sig = torch.cat([torch.sin(st), torch.cos(st)], dim=1)
    if channels % 2:
        sig = F.pad(sig, (0, 1))
    return sig.reshape(1, -1, channels)


def attention_bias(inputs, mode):
    """func.py:372-400."""
    inf = Cfg.inf
    if mode == "causal":
        length = inputs
        lt = torch.tril(torch.ones(length, length, dtype=torch.get_default_dtype()))
        return (-inf * (1.0 - lt)).reshape(1, 1, length, length)
    if mode == "masking":
        return ((1.0 - inputs) * -inf)[:, None, None, :]
    if mode == "aan":
        length = inputs.shape[1]
        cum = torch.cumsum(torch.eye(length, dtype=inputs.dtype), dim=0)[None]
        mask = inputs[:, None, :] * inputs[:, :, None]
        mask = mask * cum
        weight = torch.softmax(mask + (1.0 - mask) * -inf, dim=-1)
        return weight * mask
    raise ValueError(mode)


def remove_invalid_seq(sequence, mask):
    """util.py:274-287: drop all-pad columns, always keep column 0."""
    col = mask.sum(0)
    col[0] = col[0] + 1.0
    keep = col != 0
    return sequence[:, keep], mask[:, keep]


def label_smooth(labels, vocab_size, factor, dtype):
    """util.py:88-103."""
    flat = labels.reshape(-1)
    if 0.0 < factor < 1.0:
        n = float(vocab_size - 1)
        p = 1.0 - factor
        q = factor / n
        t = torch.full((flat.shape[0], vocab_size), q, dtype=dtype)
        t[torch.arange(flat.shape[0]), flat] = p
        # fp32 arithmetic like the reference's tf.float32 constants
        normalizing = -(np.float32(p) * np.log(np.float32(p)) +
                        np.float32(n) * np.float32(q) * np.log(np.float32(q) + np.float32(1e-20)))
        normalizing = float(normalizing)
    else:
        t = torch.zeros((flat.shape[0], vocab_size), dtype=dtype)
        t[torch.arange(flat.shape[0]), flat] = 1.0
        normalizing = 0.0
    return t, normalizing


# --------------------------------------------------------------------------
# models restated
# --------------------------------------------------------------------------
def _emb_name(hp, which):
    if hp.shared_source_target_embedding:
        return "embedding"
    if which == "src":
        return "src_embedding"
    if which == "tgt":
        return "tgt_embedding"
    return "tgt_embedding" if hp.shared_target_softmax_embedding else "softmax_embedding"


def encoder(source, hp, P, model_name, training=True):
    """transformer.py:15-84 (== transformer_aan.py:20-89 except for the
    decoder_initializer entries; rpr flags transformer_rpr.py:54-55)."""
    dt = P["bias"].dtype
    H = hp.hidden_size
    mask = (source != 0).to(dt)
    source, mask = remove_invalid_seq(source, mask)
    x = _st_fwd(P[_emb_name(hp, "src")])[source] * (H ** 0.5)
    x = x + P["bias"]
    x = x + timing_signal(x.shape[1], x.shape[2], dt)
    x = _st(dropout(x, hp.dropout, training), "embed")
    rpr = model_name == "transformer_rpr"
    for l in range(hp.num_encoder_layer):
        pre = "encoder/layer_%d" % l
        y = dot_attention(x, None, attention_bias(mask, "masking"), H, P,
                          pre + "/self_attention", hp.num_heads,
                          drop=hp.attention_dropout, use_rpr=rpr,
                          max_rel=hp.max_relative_position, training=training)['output']
        x = layer_norm(residual_fn(x, y, hp.residual_dropout, training), P, pre + "/self_attention")
        y = ffn_layer(x, P, pre + "/feed_forward", hp.relu_dropout, training)
        x = layer_norm(residual_fn(x, y, hp.residual_dropout, training), P, pre + "/feed_forward")
    B = x.shape[0]
    if model_name in ("transformer_aan", "transformer_fuse"):
        dec_init = {"layer_%d" % l: {"aan": torch.zeros(B, 1, H, dtype=dt)}
                    for l in range(hp.num_decoder_layer)}
    else:
        dec_init = {"layer_%d" % l: {"k": torch.zeros(B, 0, H, dtype=dt),
                                     "v": torch.zeros(B, 0, H, dtype=dt)}
                    for l in range(hp.num_decoder_layer)}
    return {"encodes": x, "decoder_initializer": dec_init, "mask": mask}


def average_attention(x, mask, state, layer, hp, is_training):
    """transformer_aan.py:92-117."""
    if is_training:
        if hp.aan_mask:
            return torch.matmul(attention_bias(mask, "aan"), x)
        b = torch.cumsum(mask, dim=1)
        b = torch.where(b <= 0., torch.ones_like(b), b)[:, :, None]
        return torch.cumsum(x, dim=1) / b
    cache = state['decoder']['state']['layer_%d' % layer]
    x_fwd = (x + cache['aan']) / float(state['time'] + 1)
    cache['aan'] = x + cache['aan']
    return x_fwd


def decoder(target, state, hp, P, model_name, training=True):
    """transformer.py:87-218 / transformer_aan.py:120-260 /
    transformer_rpr.py (same + use_relative_pos)."""
    dt = P["bias"].dtype
    H = hp.hidden_size
    mask = (target != 0).to(dt)
    is_training = ('decoder' not in state)
    if is_training:
        target, mask = remove_invalid_seq(target, mask)
    inputs = _st_fwd(P[_emb_name(hp, "tgt")])[target] * (H ** 0.5)
    inputs = inputs + P["bias"]
    if is_training:
        inputs = F.pad(inputs, (0, 0, 1, 0))[:, :-1, :]
        inputs = inputs + timing_signal(inputs.shape[1], inputs.shape[2], dt)
    else:
        if bool((target == hp.tgt_vocab.pad()).all()):
            inputs = torch.zeros_like(inputs)
        mask = torch.ones_like(mask)
        inputs = inputs + timing_signal(1, inputs.shape[2], dt, time=state['time'])
    x = _st(dropout(inputs, hp.dropout, training), "embed")
    rpr = model_name == "transformer_rpr"
    aan = model_name == "transformer_aan"
    dstep = None if is_training else state['time']
    for l in range(hp.num_decoder_layer):
        pre = "decoder/layer_%d" % l
        lcache = None if is_training else state['decoder']['state']['layer_%d' % l]
        if model_name == "transformer_fuse":
            # transformer_fuse.py:131-160
            r = dot_attention(x, state['encodes'], attention_bias(state['mask'], "masking"), H, P,
                              pre + "/fuse_attention", hp.num_heads, cache=lcache,
                              drop=hp.attention_dropout, training=training,
                              fuse_mask=attention_bias(mask, "aan") if is_training else state['time'])
            if not is_training:
                lcache.update(r['cache'])
            x = layer_norm(residual_fn(x, r['output'], hp.residual_dropout, training), P, pre + "/fuse_attention")
            y = ffn_layer(x, P, pre + "/feed_forward", hp.relu_dropout, training)
            x = layer_norm(residual_fn(x, y, hp.residual_dropout, training), P, pre + "/feed_forward")
            continue
        if aan:
            assert [s.lower() for s in hp.strategies] == ["aan"]
            y = _st(average_attention(x, mask, state, l, hp, is_training), "aan")
            if hp.use_ffn:
                y = ffn_layer(y, P, pre + "/average_attention", hp.relu_dropout, training)
            z = linear(torch.cat([x, y], dim=-1), P, pre + "/average_attention/z_project")
            i, f = torch.split(z, H, dim=-1)
            y = _st(torch.sigmoid(i) * x + torch.sigmoid(f) * y, "aan")
            x = layer_norm(residual_fn(x, y, hp.residual_dropout, training), P, pre + "/average_attention")
        else:
            r = dot_attention(x, None, attention_bias(mask.shape[1], "causal").to(dt), H, P,
                              pre + "/self_attention", hp.num_heads, cache=lcache,
                              drop=hp.attention_dropout, use_rpr=rpr,
                              max_rel=hp.max_relative_position,
                              decode_step=dstep if rpr else None, training=training)
            if not is_training:
                lcache.update(r['cache'])
            x = layer_norm(residual_fn(x, r['output'], hp.residual_dropout, training), P, pre + "/self_attention")
        r = dot_attention(x, state['encodes'], attention_bias(state['mask'], "masking"), H, P,
                          pre + "/cross_attention", hp.num_heads, cache=lcache,
                          drop=hp.attention_dropout, use_rpr=rpr,
                          max_rel=hp.max_relative_position,
                          decode_step=dstep if rpr else None, training=training)
        if not is_training:
            lcache.update(r['cache'])
        x = layer_norm(residual_fn(x, r['output'], hp.residual_dropout, training), P, pre + "/cross_attention")
        y = ffn_layer(x, P, pre + "/feed_forward", hp.relu_dropout, training)
        x = layer_norm(residual_fn(x, y, hp.residual_dropout, training), P, pre + "/feed_forward")
    feature = x
    if 'dev_decode' in state:
        feature = x[:, -1, :]
    feature = feature.reshape(-1, hp.embed_size)
    logits = _st_bwd(torch.matmul(feature, _st_fwd(P[_emb_name(hp, "softmax")]).t()), "logits")
    logits32 = logits  # tf.cast(logits, tf.float32): the restatement already runs >= fp32
    if 'dev_decode' in state or not is_training:
        # loss tensors are built by the reference graph but never fetched on
        # the decode path (search.py:141 only uses logits, state)
        return None, logits32, state, None
    soft, normalizer = label_smooth(target, logits32.shape[-1], hp.label_smooth, logits32.dtype)
    centropy = -(soft * torch.log_softmax(logits32, dim=-1)).sum(-1)
    centropy = centropy - normalizer
    centropy = centropy.reshape(target.shape)
    per_sample = (centropy * mask).sum(-1) / mask.sum(-1)
    if target.shape[0] == 0:
        loss = torch.zeros((), dtype=logits32.dtype)
    else:
        loss = per_sample.mean()
    return loss, logits32, state, per_sample


def closing_dropout(hp):
    """util.py:106-114."""
    for k in list(hp.values().keys()):
        if 'dropout' in k:
            setattr(hp, k, 0.0)
        if 'label_smoothing' in k:
            setattr(hp, k, 0.0)
    return hp


def train_fn(features, hp, P, model_name, training=True):
    """transformer.py:221-232."""
    state = encoder(features['source'], hp, P, model_name, training)
    loss, logits, state, per_sample = decoder(features['target'], state, hp, P, model_name, training)
    return {"loss": loss, "logits": logits, "per_sample_loss": per_sample}


def score_fn(features, hp, P, model_name):
    """transformer.py:235-249."""
    hp = closing_dropout(copy.copy(hp))
    hp.label_smooth = 0.0
    state = encoder(features['source'], hp, P, model_name, False)
    _, _, _, scores = decoder(features['target'], state, hp, P, model_name, False)
    return {"score": scores}


def infer_fn(hp, P, model_name):
    """transformer.py:252-285."""
    hp = closing_dropout(copy.copy(hp))

    def encoding_fn(source):
        state = encoder(source, hp, P, model_name, False)
        state["decoder"] = {"state": state["decoder_initializer"]}
        return state

    def decoding_fn(target, state, time):
        if hp.search_mode == "cache":
            state['time'] = time
            _, step_logits, step_state, _ = decoder(target, state, hp, P, model_name, False)
            del state['time']
        else:
            estate = encoder(state, hp, P, model_name, False)
            estate['dev_decode'] = True
            _, step_logits, _, _ = decoder(target, estate, hp, P, model_name, False)
            step_state = state
        return step_logits, step_state

    return encoding_fn, decoding_fn


# --------------------------------------------------------------------------
# search.py restated
# --------------------------------------------------------------------------
def _map_structure(fn, nest):
    if isinstance(nest, dict):
        return {k: _map_structure(fn, v) for k, v in nest.items()}
    return fn(nest)


def _merge(x, axis=0):
    """util.py:154-162."""
    if x.dim() < axis + 2:
        return x
    shp = list(x.shape)
    shp[axis] *= shp[axis + 1]
    shp.pop(axis + 1)
    return x.reshape(shp)


def _unmerge(x, depth, axis=0):
    """util.py:165-173."""
    if x.dim() < axis + 1:
        return x
    shp = list(x.shape)
    width = shp[axis] // depth
    return x.reshape(shp[:axis] + [depth, width] + shp[axis + 1:])


def _dict_update(d, u):
    """util.py:117-124."""
    for k, v in u.items():
        if isinstance(v, dict):
            d[k] = _dict_update(d.get(k, {}), v)
        else:
            d[k] = v
    return d


def _top_k(x, k):
    """tf.nn.top_k: descending, ties -> lower index first."""
    xs = x.detach().cpu().numpy()
    idx = np.argsort(-xs, axis=-1, kind="stable")[..., :k]
    vals = np.take_along_axis(xs, idx, axis=-1)
    return torch.tensor(vals, dtype=x.dtype), torch.tensor(idx, dtype=torch.long)


def _gather_beams(x, idx):
    """tf.gather_nd(x, stack([batch_pos, idx])) for x [B, K, ...], idx [B, k]."""
    B = x.shape[0]
    bpos = torch.arange(B)[:, None].expand_as(idx)
    return x[bpos, idx]


def beam_search(features, encoding_fn, decoding_fn, hp):
    """search.py:19-275.  Scores / log-probs are fp32 like the reference
    (tfdtype = tf.float32, search.py:41-43); returns numpy arrays plus the
    number of decoding steps taken."""
    f32 = torch.float32
    decode_length = hp.decode_length
    beam_size = hp.beam_size
    alpha = hp.decode_alpha
    eos_id = hp.tgt_vocab.eos()
    pad_id = hp.tgt_vocab.pad()
    source = features["source"]
    batch_size = source.shape[0]
    trace = getattr(hp, "search_trace", None)      # a list to append per-step candidate tables to, or None
    if hp.search_mode == "cache":
        model_state = encoding_fn(source)
    else:
        model_state = source
    src_mask = (source != 0).to(f32)
    source_length = src_mask.sum(-1)
    max_target_length = source_length + decode_length

    def tile(x):
        x = x.unsqueeze(1)
        reps = [1] * x.dim()
        reps[1] = beam_size
        return x.repeat(*reps)
    model_state = _map_structure(tile, model_state)

    init_log_probs = torch.tensor([[0.] + [F32_MIN] * (beam_size - 1)], dtype=f32).repeat(batch_size, 1)
    init_scores = torch.zeros_like(init_log_probs)
    init_seq = torch.full((batch_size, beam_size, 1), pad_id, dtype=torch.long)
    fin_seq = torch.zeros_like(init_seq)
    fin_scores = torch.full((batch_size, beam_size), F32_MIN, dtype=f32)
    fin_flags = torch.zeros((batch_size, beam_size), dtype=torch.bool)

    if hp.search_mode == "cache":
        # search.py:56-77 cache_init: dummy step, then restore original entries
        flat_seq = _merge(init_seq, 0)
        flat_state = _map_structure(lambda x: _merge(x, 0), model_state)
        import copy as _copy
        snapshot = _copy_nest(model_state)
        _, step_state = decoding_fn(flat_seq[:, -1:], flat_state, 0)
        new_state = _map_structure(lambda x: _unmerge(x, batch_size, 0), step_state)
        model_state = _dict_update(new_state, snapshot)

    seq, log_probs, scores = init_seq, init_log_probs, init_scores
    state = model_state
    time = 0
    mtl32 = max_target_length.to(f32)
    while True:
        # ---- _not_finished (search.py:85-113)
        max_lp = torch.pow((5. + mtl32) / 6., alpha)
        best_alive = log_probs[:, 0] / max_lp
        worst_fin = (fin_scores * fin_flags.to(f32)).min(dim=1).values
        unfinish = 1. - fin_flags.any(dim=1).to(f32)
        worst_fin = worst_fin + unfinish * F32_MIN
        bound_is_met = bool((worst_fin > best_alive).all())
        length_is_met = bool((time < max_target_length.to(torch.int32)).any())
        if not ((not bound_is_met) and length_is_met):
            break
        # ---- _step_fn (search.py:115-236)
        flat_seq = _merge(seq, 0)
        flat_state = _map_structure(lambda x: _merge(x, 0), state)
        if hp.search_mode == "cache":
            decode_target = flat_seq[:, -1:]
        else:
            decode_target = F.pad(flat_seq[:, 1:], (0, 1), value=1)
        step_logits, step_state = decoding_fn(decode_target, flat_state, time)
        step_logits = step_logits.to(f32) / hp.beam_search_temperature
        step_lp = step_logits - torch.logsumexp(step_logits, dim=-1, keepdim=True)
        V = step_lp.shape[-1]
        if time < 1:
            eos_mask = (torch.arange(V) == eos_id).to(f32)
            step_lp = step_lp + eos_mask[None, :] * -Cfg.inf
        step_lp = _unmerge(step_lp, batch_size, 0)
        step_state = _map_structure(lambda x: _unmerge(x, batch_size, 0), step_state)
        curr_lp = log_probs[:, :, None] + step_lp
        length_penalty = float(np.power(np.float32((5.0 + np.float32(time + 1)) / 6.), np.float32(alpha)))
        length_penalty = torch.tensor(length_penalty, dtype=f32)
        curr_scores = curr_lp / length_penalty
        flat_scores = _merge(curr_scores, 1)
        topk_scores, topk_idx = _top_k(flat_scores, 2 * beam_size)
        if trace is not None:
            # checker aid (tests/test_gpu_fullsize.py): the 2K candidates search.py:172-176 keeps PLUS the runner-ups
            # (round 6: eight of them, one before -- a candidate the other path ranks inside its 2K was sometimes below
            # the single runner-up and had no gap on record), so that a path that leaves this one at some step can be
            # judged by the score gap it had to bridge there
            ts, ti = _top_k(flat_scores, min(2 * beam_size + TRACE_RUNNER_UPS, flat_scores.shape[1]))
            trace.append((ts.numpy().copy(), ti.numpy().copy()))
        beam_idx = topk_idx // V
        sym_idx = topk_idx % V
        curr_seq = _gather_beams(seq, beam_idx)
        curr_seq = torch.cat([curr_seq, sym_idx[:, :, None]], dim=2)
        over = (time >= max_target_length.to(torch.int32))[:, None]
        curr_fin = (sym_idx == eos_id) | over
        alive_scores = topk_scores + curr_fin.to(f32) * F32_MIN
        alive_scores, alive_idx = _top_k(alive_scores, beam_size)
        alive_seq = _gather_beams(curr_seq, alive_idx)
        alive_beam = _gather_beams(beam_idx, alive_idx)
        alive_state = _map_structure(lambda x: _gather_beams(x, alive_beam), step_state)
        alive_lp = alive_scores * length_penalty
        cfs = topk_scores + (1.0 - curr_fin.to(f32)) * F32_MIN
        all_flags = torch.cat([fin_flags, curr_fin], dim=1)
        all_scores = torch.cat([fin_scores, cfs], dim=1)
        new_fin_scores, fin_idx = _top_k(all_scores, beam_size)
        new_fin_flags = _gather_beams(all_flags, fin_idx)
        pad_seq = torch.full((batch_size, beam_size, 1), pad_id, dtype=torch.long)
        all_seq = torch.cat([torch.cat([fin_seq, pad_seq], dim=2), curr_seq], dim=1)
        fin_seq = _gather_beams(all_seq, fin_idx)
        fin_scores, fin_flags = new_fin_scores, new_fin_flags
        seq, log_probs, scores, state = alive_seq, alive_lp, alive_scores, alive_state
        time += 1

    any_fin = fin_flags.any(dim=1)
    final_seqs = torch.where(any_fin[:, None, None], fin_seq, seq)
    final_scores = torch.where(any_fin[:, None], fin_scores, scores)
    return {"seq": final_seqs[:, :, 1:].numpy(), "score": final_scores.numpy(), "steps": time}


def _copy_nest(n):
    if isinstance(n, dict):
        return {k: _copy_nest(v) for k, v in n.items()}
    return n


def decode_hypothesis(seqs, hp):
    """evalu.py:14-46: beam 0, cut at first eos or pad."""
    out = []
    for b in range(seqs.shape[0]):
        ids = []
        for t in seqs[b, 0]:
            t = int(t)
            if t == hp.tgt_vocab.eos() or t == hp.tgt_vocab.pad():
                break
            ids.append(t)
        out.append(ids)
    return out


# --------------------------------------------------------------------------
# utils/cycle.py + tf.train.AdamOptimizer + lrs/noamlr.py restated
# --------------------------------------------------------------------------
def noam_lr(step, hp):
    """lrs/noamlr.py:27-34 + lrs/lr.py:43-45 (clamped)."""
    step = float(step)
    w = float(hp.warmup_steps)
    decay = float(hp.hidden_size) ** -0.5 * min((step + 1) * (w ** -1.5), (step + 1) ** -0.5)
    lr = hp.lrate * decay
    return max(min(lr, hp.max_lrate), hp.min_lrate)


def global_norm(tensors):
    return torch.sqrt(sum((t.double() ** 2).sum() for t in tensors)).to(tensors[0].dtype)


def adam_step(P, G, M, V, t, lr, hp):
    """TF1 AdamOptimizer (main.py:178-181): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; theta -= lr_t * m / (sqrt(v) + eps)  (eps outside sqrt).
    cycle.py:94-101: gnorm/pnorm and clip iff clip_grad_norm is a non-zero
    float.  ``t`` is the 1-based update count."""
    names = list(P.keys())
    gnorm = global_norm([G[n] for n in names])
    pnorm = global_norm([P[n] for n in names])
    clip = hp.clip_grad_norm or None
    scale = 1.0
    if isinstance(clip, float):
        scale = clip / max(float(gnorm), clip)
    b1, b2, eps = hp.beta1, hp.beta2, hp.epsilon
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    with torch.no_grad():
        for n in names:
            g = G[n] * scale
            M[n].mul_(b1).add_(g, alpha=1 - b1)
            V[n].mul_(b2).addcmul_(g, g, value=1 - b2)
            P[n].sub_(lr_t * M[n] / (V[n].sqrt() + eps))
    return float(gnorm), float(pnorm)


def train_step(P, M, V, features, hp, model_name, step, training=True):
    """One update of main.py:268-332 with update_cycle=1, one tower:
    loss -> grads -> (clip) -> Adam with lr = noam(step)."""
    for p in P.values():
        p.requires_grad_(True)
        p.grad = None
    out = train_fn(features, hp, P, model_name, training)
    out["loss"].backward()
    G = OrderedDict((n, p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in P.items())
    lr = noam_lr(step, hp) if hp.lrate_strategy == "noam" else hp.lrate
    for p in P.values():
        p.requires_grad_(False)
    gnorm, pnorm = adam_step(P, G, M, V, step + 1, lr, hp)
    return float(out["loss"].detach()), gnorm, pnorm, G
