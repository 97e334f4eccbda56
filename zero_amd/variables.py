# coding: utf-8
"""Variable store of one model scope -- the HBM layout of the parameters.

The reference keeps parameters in TF's variable store under ``{scope_name}/…``
(transformer.py:222; names from func.py:48,58,196,207-212,278,297-298 and
modules/rpr.py:56), stored fp32 and cast to the compute dtype on read
(utils/dtype.py:55-69).  Here every trainable variable of the scope lives in
ONE flat fp32 buffer (master weights) with three same-shaped companions:

    master fp32 | grad fp32 | adam m fp32 | adam v fp32 | shadow bf16

so that gradient all-reduce, global-norm, clipping and Adam are each a single
pass over contiguous HBM (utils/cycle.py:94-101 semantics), and the bf16
shadow the MFMA GEMMs read is refreshed by the Adam kernel itself.  Variable
names and logical shapes equal the reference's, so checkpoints can be
exchanged by name (``export`` / ``load``).

Embedding tables are physically padded to a multiple of 8 rows (zero rows,
zero gradients) so that vocabulary-sized GEMM dimensions stay 16-byte aligned.
"""

import math
from collections import OrderedDict

import numpy as np
import torch

ALIGN = 64  # elements; keeps every variable 256-byte (fp32) / 128-byte (bf16) aligned


def _attn_specs(prefix, H, self_att, rpr, nrel, d):
    p = prefix + "/dot_attention/"
    v = []
    if self_att:
        v += [(p + "qkv_map/W_0_0", (H, 3 * H), "w"), (p + "qkv_map/b_0", (3 * H,), "zeros")]
    else:
        for m in ("q_map", "k_map", "v_map"):
            v += [(p + m + "/W_0_0", (H, H), "w"), (p + m + "/b_0", (H,), "zeros")]
    if rpr:
        v += [(p + "rpr_keys/embeddings", (nrel, d), "w"),
              (p + "rpr_values/embeddings", (nrel, d), "w")]
    v += [(p + "o_map/W_0_0", (H, H), "w"), (p + "o_map/b_0", (H,), "zeros"),
          (prefix + "/layer_norm/scale", (H,), "ones"), (prefix + "/layer_norm/offset", (H,), "zeros")]
    return v


def _ffn_specs(prefix, H, F, with_ln=True):
    p = prefix + "/ffn_layer/"
    v = [(p + "enlarge/W_0_0", (H, F), "w"), (p + "enlarge/b_0", (F,), "zeros"),
         (p + "output/W_0_0", (F, H), "w"), (p + "output/b_0", (H,), "zeros")]
    if with_ln:
        v += [(prefix + "/layer_norm/scale", (H,), "ones"), (prefix + "/layer_norm/offset", (H,), "zeros")]
    return v


def variable_specs(params, model_name):
    """[(name, logical_shape, kind, layer)] in the reference's creation order
    (transformer.py:16-33,88-102,184-192; transformer_aan.py:165-192;
    transformer_rpr.py:54-55,144-146,167-169; transformer_fuse.py:131-160)."""
    H, E, F = params.hidden_size, params.embed_size, params.filter_size
    if H != E:
        raise ValueError("hidden_size must equal embed_size for the Transformer models "
                         "(embeddings are added to H-wide layers, transformer.py:29-31)")
    d = H // params.num_heads
    rpr = model_name == "transformer_rpr"
    aan = model_name == "transformer_aan"
    fuse = model_name == "transformer_fuse"
    nrel = 2 * params.max_relative_position + 1
    Vs, Vt = params.src_vocab.size(), params.tgt_vocab.size()
    shared = params.shared_source_target_embedding
    specs = [("embedding" if shared else "src_embedding", (Vs, E), "embed", None),
             ("bias", (E,), "w", None)]
    for l in range(params.num_encoder_layer):
        pre = "encoder/layer_%d" % l
        specs += [(n, s, k, l) for n, s, k in _attn_specs(pre + "/self_attention", H, True, rpr, nrel, d)]
        specs += [(n, s, k, l) for n, s, k in _ffn_specs(pre + "/feed_forward", H, F)]
    if not shared:
        specs.append(("tgt_embedding", (Vt, E), "embed", None))
    for l in range(params.num_decoder_layer):
        pre = "decoder/layer_%d" % l
        if fuse:      # transformer_fuse.py:131-160: merged attention sub-layer + FFN
            specs += [(n, s, k, l) for n, s, k in _attn_specs(pre + "/fuse_attention", H, False, False, nrel, d)]
            specs += [(n, s, k, l) for n, s, k in _ffn_specs(pre + "/feed_forward", H, F)]
            continue
        if aan:
            a = pre + "/average_attention"
            if params.use_ffn:
                specs += [(n, s, k, l) for n, s, k in _ffn_specs(a, H, F, with_ln=False)]
            specs += [(a + "/z_project/W_0_0", (2 * H, 2 * H), "w", l),
                      (a + "/z_project/b_0", (2 * H,), "zeros", l),
                      (a + "/layer_norm/scale", (H,), "ones", l),
                      (a + "/layer_norm/offset", (H,), "zeros", l)]
        else:
            specs += [(n, s, k, l) for n, s, k in _attn_specs(pre + "/self_attention", H, True, rpr, nrel, d)]
        specs += [(n, s, k, l) for n, s, k in _attn_specs(pre + "/cross_attention", H, False, rpr, nrel, d)]
        specs += [(n, s, k, l) for n, s, k in _ffn_specs(pre + "/feed_forward", H, F)]
    if not shared and not params.shared_target_softmax_embedding:
        specs.append(("softmax_embedding", (Vt, E), "embed", None))
    return specs


def _scope_init(rng, shape, kind, gain):
    """modules/initializer.py:11-32 with TF1 fan rules."""
    fi, fo = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
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
    lim = math.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=shape)


def initial_values(params, model_name, seed):
    """Host-side draw from the reference's initial distribution
    (transformer.py:18,90 embeddings N(0,H^-0.5); main.py:26 scope initializer;
    transformer.py:38-44 deep_transformer_init)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape, kind, layer in variable_specs(params, model_name):
        if kind == "embed":
            v = rng.normal(0.0, params.hidden_size ** -0.5, size=shape)
        elif kind == "zeros":
            v = np.zeros(shape)
        elif kind == "ones":
            v = np.ones(shape)
        elif layer is not None and params.deep_transformer_init:
            v = _scope_init(rng, shape, "uniform_unit_scaling",
                            params.initializer_gain * (layer + 1) ** -0.5)
        else:
            v = _scope_init(rng, shape, params.initializer, params.initializer_gain)
        out[name] = v.astype(np.float32)
    return out


class VariableStore(object):
    """Flat parameter / gradient / optimiser-state buffers of one scope."""

    def __init__(self, params, model_name, device):
        self.model_name = model_name
        self.device = torch.device(device)
        self.specs = variable_specs(params, model_name)
        self.offsets, self.lshape, self.pshape = OrderedDict(), {}, {}
        off = 0
        for name, shape, kind, _ in self.specs:
            pshape = tuple(shape)
            if kind == "embed" or name.endswith("/embeddings"):
                # (relative-position tables too: their zero rows make them GEMM operands with an 8-aligned
                # contraction length as they are -- no padded copy per attention call)
                pshape = ((shape[0] + 7) // 8 * 8, shape[1])
            n = int(np.prod(pshape))
            self.offsets[name] = off
            self.lshape[name] = tuple(shape)
            self.pshape[name] = pshape
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.numel = off
        self.logical_numel = sum(int(np.prod(s)) for s in self.lshape.values())
        dev = self.device
        self.master = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        self.accum = None     # update_cycle slots (utils/cycle.py:27-36), created on demand
        self.step = 0         # number of applied updates (Adam's t)
        self.shadow_loads = 0  # refresh_shadow() calls; (shadow_loads, step) identifies the bf16 weights ("weight version")

    # -- views --------------------------------------------------------------
    def _view(self, flat, name):
        o = self.offsets[name]
        ps = self.pshape[name]
        return flat[o:o + int(np.prod(ps))].view(*ps)

    def w(self, name):
        """fp32 master view (physical shape)."""
        return self._view(self.master, name)

    def s(self, name):
        """bf16 shadow view (physical shape) -- what the GEMMs read."""
        return self._view(self.shadow, name)

    def g(self, name):
        """fp32 gradient view (physical shape)."""
        return self._view(self.grad, name)

    def names(self):
        return list(self.offsets.keys())

    # -- host <-> device -----------------------------------------------------
    def load(self, values):
        """values: {name: array (logical shape)}; refreshes the bf16 shadow."""
        for name in self.offsets:
            if name not in values:
                raise KeyError("missing variable %s" % name)
            v = torch.as_tensor(np.asarray(values[name], dtype=np.float32))
            if tuple(v.shape) != self.lshape[name]:
                raise ValueError("shape mismatch for %s: %s vs %s" % (name, tuple(v.shape), self.lshape[name]))
            dst = self.w(name)
            dst.zero_()
            if v.dim() == 2:
                dst[:v.shape[0], :v.shape[1]].copy_(v)
            else:
                dst.copy_(v)
        self.refresh_shadow()

    @property
    def weight_version(self):
        """Changes whenever the bf16 weights the kernels read may have changed: a load / EMA swap (refresh_shadow) or an
        optimiser update (utils/cycle.py advances ``step``).  Derived copies of weights are cached against it."""
        return (self.shadow_loads, self.step)

    def refresh_shadow(self):
        self.shadow_loads += 1
        if self.device.type == "cuda":
            from zero_amd import hip
            hip.lib().call("zk_cast_f32_bf16", self.master.data_ptr(), self.shadow.data_ptr(), self.numel,
                           torch.cuda.current_stream().cuda_stream)
        else:  # layout / bookkeeping tests on CPU only -- never on the compute path
            self.shadow.copy_(self.master.to(torch.bfloat16))

    def export(self, which="master"):
        """{name: np.ndarray (logical shape)} of master / grad / m / v (or of any flat buffer laid out
        like them, e.g. the EMA shadows)."""
        flat = getattr(self, which) if isinstance(which, str) else which
        out = OrderedDict()
        for name in self.offsets:
            t = self._view(flat, name)
            ls = self.lshape[name]
            if len(ls) == 2:
                t = t[:ls[0], :ls[1]]
            out[name] = t.detach().float().cpu().numpy().copy()
        return out


_STORES = {}


def get_store(params, model_name, device=None, initializer_seed=None, create=True):
    """The variable store of ``params.scope_name`` (tf.variable_scope(...,
    reuse=AUTO_REUSE) of transformer.py:222-226: first use creates, later uses share)."""
    scope = params.scope_name or "model"
    key = (scope, model_name)
    if key not in _STORES:
        if not create:
            raise KeyError("no variables for scope %s" % scope)
        if device is None:
            device = "cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
        st = VariableStore(params, model_name, device)
        seed = params.random_seed if initializer_seed is None else initializer_seed
        st.load(initial_values(params, model_name, seed))
        _STORES[key] = st
    return _STORES[key]


def reset_stores():
    _STORES.clear()
