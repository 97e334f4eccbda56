# coding: utf-8
"""Builds the (train_fn, score_fn, infer_fn) triple of one Transformer variant.

Signatures follow the reference (models/transformer.py:221-285):

    train_fn(features, params, initializer=None) -> {"loss": fp32 scalar tensor, ...}
    score_fn(features, params, initializer=None) -> {"score": fp32 [B]}
    infer_fn(params) -> (encoding_fn(source) -> state,
                         decoding_fn(target, state, time) -> (logits fp32 [B*K, V], state))

``features`` = {"source": int [B, Ls], "target": int [B, Lt]} (0 = pad).  Because there
is no autodiff engine, ``train_fn`` also runs the hand-written backward and returns the
flat gradient buffer (``"gradient"``) and the variable store (``"store"``) next to the
loss, for the build's ``tower_train_graph`` counterpart (zero_amd/main.py).
``initializer`` may be ``None`` (draw from params.initializer, main.py:26) or a
``{name: array}`` dict used when the scope's variables are first created.
"""

import copy
import threading

import torch

from zero_amd.models._core import TransformerCore
from zero_amd.variables import get_store

_CORES = {}


def closing_dropout(params):
    """utils/util.py:106-114: zero every hparam whose name contains 'dropout'."""
    for k in list(params.values().keys()):
        if 'dropout' in k:
            setattr(params, k, 0.0)
        if 'label_smoothing' in k:
            setattr(params, k, 0.0)
    return params


_LANE = threading.local()
_CORES_LOCK = threading.Lock()


class lane(object):
    """``with lane(i):`` -- the calling THREAD works on execution lane i of its model: get_core() then hands out a
    TransformerCore with its own engine (HIP stream, scratch and cache buffers, captured graphs) that SHARES the
    variable store of lane 0.  Lanes exist so that several independent decode batches can be in flight at once
    (zero_amd.evalu.decode_many: a 128-row decode step is a latency chain that leaves most of the 256 CUs idle);
    training always runs on lane 0."""

    def __init__(self, idx):
        self.idx = int(idx)

    def __enter__(self):
        self.prev = getattr(_LANE, "idx", 0)
        _LANE.idx = self.idx
        return self

    def __exit__(self, *a):
        _LANE.idx = self.prev


def current_lane():
    return getattr(_LANE, "idx", 0)


def get_core(params, model_name, initializer=None):
    """One TransformerCore per (scope, device, lane): AUTO_REUSE of transformer.py:222-226."""
    dev = "cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
    ln = current_lane()
    key = (params.scope_name or "model", model_name, dev) + ((ln,) if ln else ())
    core = _CORES.get(key)
    if core is None:
        with _CORES_LOCK:
            core = _CORES.get(key)
            if core is None:
                store = get_store(params, model_name, dev)
                if isinstance(initializer, dict):
                    store.load(initializer)
                core = TransformerCore(params, model_name, store, dev)
                _CORES[key] = core
    else:
        core.hp = params
    return core


def reset_cores():
    from zero_amd.variables import reset_stores
    _CORES.clear()
    reset_stores()


def _n_sentences(features):
    if "B" in features:
        return features["B"]
    return int(features["source"].shape[0])


def build(model_name):
    def train_fn(features, params, initializer=None, on_ready=None):
        core = get_core(params, model_name, initializer)
        if _n_sentences(features) == 0:
            # transformer.py:213-216: a tower without sentences contributes loss 0 (and zero gradients);
            # it still hands every bucket to the all-reduce so that the other ranks are not left waiting
            from zero_amd.utils.parallel import layer_buckets
            core.eng.zero(core.store.grad)
            if on_ready is not None:
                for key in layer_buckets(core.store)[1]:
                    on_ready(key)
            zero = torch.zeros(1, dtype=torch.float32, device=core.store.device)
            return {"loss": zero[0], "gradient": core.store.grad, "store": core.store,
                    "per_sample_loss": zero[:0]}
        batch = features if "B" in features else core.upload(features["source"], features["target"])
        loss, per_sample, _ = core.forward(batch, train=True, save=True)
        core.backward(on_ready)
        return {"loss": loss[0], "gradient": core.store.grad, "store": core.store,
                "per_sample_loss": per_sample}

    def score_fn(features, params, initializer=None):
        params = copy.copy(params)
        params = closing_dropout(params)
        params.label_smooth = 0.0
        core = get_core(params, model_name, initializer)
        if _n_sentences(features) == 0:
            return {"score": torch.zeros(0, dtype=torch.float32, device=core.store.device)}
        batch = features if "B" in features else core.upload(features["source"], features["target"])
        _, per_sample, _ = core.forward(batch, train=False, save=False, label_smooth=0.0)
        return {"score": per_sample}

    def infer_fn(params):
        params = copy.copy(params)
        params = closing_dropout(params)
        from zero_amd.models._decode import make_infer_fns
        return make_infer_fns(params, model_name)

    return train_fn, score_fn, infer_fn
