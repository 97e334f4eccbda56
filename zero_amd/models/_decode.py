# coding: utf-8
"""Incremental (cached) and full-recompute decoding steps on HIP kernels.

``infer_fn`` of the reference (models/transformer.py:252-285, transformer_aan.py:294-327)
returns two closures; so does :func:`make_infer_fns`:

  encoding_fn(source)                -> state
  decoding_fn(target, state, time)   -> (logits fp32 [B*K, ld>=V], state)

MI355X layout of ``state`` (instead of the reference's beam-tiled nest, search.py:36-39):
  * beam-invariant tensors -- encoder output, source mask and the cross-attention keys /
    values ``mk`` / ``mv`` of every layer (func.py:206-216) -- are stored ONCE per sentence
    and never tiled or reordered; the attention kernel maps query row b*K+k to sentence b
    (``kv_group=K``);
  * per-beam caches -- self-attention ``k`` / ``v`` (func.py:199-205) as preallocated
    [B*K, Tmax, H] bf16 buffers written in place at position ``time`` (no concat), the AAN
    running sum (transformer_aan.py:110-112) as fp32 [B*K, H] -- are double-buffered and
    reordered by a row gather (search.py:206-209).
``cache_init``'s dummy step (search.py:56-77) is unnecessary: mk/mv are computed by
``encoding_fn``; the values are the same.

search_mode="dev" (transformer.py:277-281) recomputes the encoder and the whole prefix
through the training-path decoder each step.
"""

import numpy as np
import torch

from zero_amd.func import Mat
from zero_amd.models._factory import get_core
from zero_amd.models import _decode_f32 as _f32
from zero_amd.utils import dtype as zdtype

F32 = torch.float32

# Start-up of a decode batch = the part that allocates (engine buffers, pinned staging) and captures the two step
# graphs.  With several batches in flight on execution lanes (evalu.decode_many, one host thread per lane) the
# start-ups are serialised by this lock: an allocation or a host-memory pin in one thread while another thread is
# inside a stream capture invalidated that capture on ROCm 7.x ("operation failed due to a previous error during
# capture"), whereas a thread that only replays graphs beside a capturing one is fine.  The replay loop -- ~95 % of a
# batch's time -- runs unlocked; a single-threaded caller never waits.
import threading as _threading
STARTUP_LOCK = _threading.RLock()


def startup_begin():
    STARTUP_LOCK.acquire()


def startup_end(state):
    """Release the start-up lock of this batch (idempotent)."""
    if state is not None and state.get("_startup_held"):
        state["_startup_held"] = False
        STARTUP_LOCK.release()


def _startup_settled(state):
    g = state.get("graphs")
    return bool(g) and len(g) >= 2 and all(not isinstance(v, str) for v in g.values())


# ---- decode-step graphs live across batches ----------------------------------------------------------------------
# Every launch argument of a step graph is either a per-step value read from device memory or a property of the batch
# SHAPE (beam rows, padded source length, cache length) and of named engine buffers, whose addresses are stable once
# they have reached their largest size.  So the two parity graphs of a batch are kept per shape (per core, i.e. per
# execution lane) and a later batch of the same shape replays them instead of capturing its own: ~2.5 ms of the
# ~4.7 ms a batch spends before its first replayed step.  An entry is only reused if the engine has not replaced a
# buffer since (realloc_gen) AND the data pointers the graph was captured with are the ones the new batch uses.
STEP_GRAPH_CACHE = 24        # shapes per core; least recently used goes first


def _graph_pointers(state, book):
    core = state["_core"]
    e = core.eng
    ptrs = [state["pack_dev"].data_ptr(), state["out_dev"].data_ptr(), state["mask"].data_ptr(),
            state["encodes"].ptr, e.seed.data_ptr(), core.store.shadow.data_ptr()]
    for l in range(core.hp.num_decoder_layer):
        lay = state["decoder"]["state"]["layer_%d" % l]
        ptrs += [lay["mk"].ptr, lay["mv"].ptr]
    ptrs.append(core.store.master.data_ptr())
    for nm in ("dc.aan.0", "dc.aan.1", "dc.k.0", "dc.k.1", "dc.v.0", "dc.v.1",
               "dq.aan.0", "dq.aan.1", "dq.k.0", "dq.k.1", "dq.v.0", "dq.v.1", "dq.logits"):
        b = e.bufs.get(nm)
        ptrs.append(b.data_ptr() if b is not None else 0)
    ptrs += [m.ptr for _, m in sorted(state.get("wt", {}).items())]
    return tuple(ptrs) + (tuple(book) if book is not None else ())


def _destroy_graphs(core, graphs):
    for g in graphs.values():
        if not isinstance(g, str):
            core.eng.lib.call("zk_graph_destroy", g)
    graphs.clear()


def adopt_graphs(state, book, temperature, forbid_value, noise):
    """First step of a batch: take over the cached parity graphs of its shape, if they are still valid."""
    import collections
    import os
    core = state["_core"]
    e = core.eng
    # (zk_dec_group(-1) = the rows-per-workgroup geometry in force: graphs captured with several batches in flight --
    # 16 rows per workgroup -- must not be adopted by a single-stream decode, nor the reverse)
    from zero_amd import hip as _hip
    key = (state["B"], state["K"], state["Ls"], state["Tmax"], book is not None, float(temperature),
           float(forbid_value), bool(noise), int(_hip.lib().raw("zk_dec_group")(-1)), bool(state.get("f32")))
    state["_gkey"] = key
    if os.environ.get("ZERO_HIP_DECODE_GRAPH_CACHE", "1") == "0":
        return
    cache = core.__dict__.setdefault("_step_graph_cache", collections.OrderedDict())
    ent = cache.pop(key, None)
    if ent is None:
        return
    if ent["gen"] != e.realloc_gen or ent["ptrs"] != _graph_pointers(state, book):
        torch.cuda.current_stream(e.device).synchronize()
        _destroy_graphs(core, ent["graphs"])
        return
    state["graphs"] = ent["graphs"]
    core._graph_adoptions = core.__dict__.get("_graph_adoptions", 0) + 1


def retire_graphs(state):
    """End of a batch: complete parity graphs go to the shape cache of their core, everything else is destroyed."""
    import collections
    import os
    graphs = state.get("graphs") if hasattr(state, "get") else None
    if not graphs:
        return
    core = state["_core"]
    e = core.eng
    complete = len(graphs) >= 2 and all(not isinstance(g, str) for g in graphs.values())
    if complete:
        # a whole batch ran from captured graphs: the step's scratch exists for this many beam rows
        core._decode_warm_rows = max(core.__dict__.get("_decode_warm_rows", 0), state["BK"])
    key = state.get("_gkey")
    torch.cuda.current_stream(e.device).synchronize()
    if complete and key is not None and os.environ.get("ZERO_HIP_DECODE_GRAPH_CACHE", "1") != "0":
        cache = core.__dict__.setdefault("_step_graph_cache", collections.OrderedDict())
        old = cache.pop(key, None)
        if old is not None and old["graphs"] is not graphs:
            _destroy_graphs(core, old["graphs"])
        cache[key] = {"gen": e.realloc_gen, "ptrs": _graph_pointers(state, state.get("book")), "graphs": dict(graphs)}
        while len(cache) > STEP_GRAPH_CACHE:
            _, ev = cache.popitem(last=False)
            _destroy_graphs(core, ev["graphs"])
        state["graphs"] = {}
        return
    _destroy_graphs(core, graphs)


class lanes_mode(object):
    """Kernel geometry for ``n`` decode batches in flight (evalu.decode_many).  Alone, the fused attention launch is
    fastest with one sentence (4 beam rows) per workgroup: 256 workgroups of 121 KB LDS, i.e. the whole chip, 12.7 us
    against 16.5 us with 16 rows per workgroup.  With several batches in flight that launch is the one kernel of a step
    the lanes cannot overlap (it owns every CU's LDS); at 16 rows per workgroup it covers a quarter of the chip and the
    lanes' launches run side by side (measured: 4 lanes 1.5-1.6 k -> 1.9-2.2 k sentences/s)."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        import os
        from zero_amd import hip
        rows = 16 if self.n > 1 else 0
        self.prev = hip.lib().raw("zk_dec_group")(rows)
        return self

    def __exit__(self, *a):
        from zero_amd import hip
        hip.lib().raw("zk_dec_group")(self.prev)


class DecodeState(dict):
    """Nested-dict state with the cache plumbing the search needs."""

    def reorder(self, index_dev, time_dev=None, defer_aan=False):
        """Gather every per-beam cache by flat beam index [B*K] (device int32).  time_dev: the
        number of filled cache slots lives in device memory (hipGraph replay).  The caches of all
        layers are slabs of one buffer, so each kind (aan / k / v) is ONE launch."""
        core = self["_core"]
        e = core.eng
        BK, H, t = self["BK"], core.H, self["time_filled"]
        nl = core.hp.num_decoder_layer
        pp = self["_pp"]
        lay0 = self["decoder"]["state"]["layer_0"]
        if "aan" in lay0:
            src = e.buf("dc.aan.%d" % pp, (nl, BK, H), F32)
            dst = e.buf("dc.aan.%d" % (1 - pp), (nl, BK, H), F32)
            if defer_aan:
                # the decoder-input launch of the step does it (zk_dec_embed: gather_src / gather_idx)
                self["_aan_gather"] = (src, index_dev)
            else:
                e.lib.call("zk_gather_rows_ex", src.data_ptr(), H * 4, index_dev.data_ptr(), dst.data_ptr(), H * 4,
                           nl * BK, H * 4, BK, e.stream)
        if "k" in lay0:
            Tmax = self["Tmax"]
            for nm in ("k", "v"):
                src = e.buf("dc.%s.%d" % (nm, pp), (nl, BK, Tmax, H))
                dst = e.buf("dc.%s.%d" % (nm, 1 - pp), (nl, BK, Tmax, H))
                if time_dev is not None:
                    e.lib.call("zk_cache_rows", src.data_ptr(), Tmax * H * 2, index_dev.data_ptr(), dst.data_ptr(),
                               Tmax * H * 2, nl * BK, H * 2, Tmax, time_dev.data_ptr(), 1, BK, e.stream)
                else:
                    e.lib.call("zk_gather_rows_ex", src.data_ptr(), Tmax * H * 2, index_dev.data_ptr(),
                               dst.data_ptr(), Tmax * H * 2, nl * BK, t * H * 2, BK, e.stream)
        self["_pp"] = 1 - pp
        self.bind_caches()

    def bind_caches(self):
        """Point every layer's cache entries at the current ping-pong half."""
        core = self["_core"]
        e = core.eng
        BK, H, nl, pp = self["BK"], core.H, core.hp.num_decoder_layer, self["_pp"]
        for l in range(nl):
            lay = self["decoder"]["state"]["layer_%d" % l]
            if "aan" in lay:
                lay["aan"] = e.buf("dc.aan.%d" % pp, (nl, BK, H), F32)[l]
            if "k" in lay:
                for nm in ("k", "v"):
                    lay[nm] = e.buf("dc.%s.%d" % (nm, pp), (nl, BK, self["Tmax"], H))[l]


DecodeStateF32 = _f32.make_state_class(DecodeState)      # decode_dtype = float32: 4-byte caches, `dq.*` buffers


def _fuse_att_ok(core, hp, K):
    """The attention sub-layers of a cached decode step as one launch per (sentence, head) (zk_dec_cross / zk_dec_self)."""
    import os
    return (os.environ.get("ZERO_HIP_DECODE_FUSE_ATT", "1") != "0" and core.d == 64
            and core.H in (128, 256, 512, 1024, 2048) and not core.fuse
            and (not core.rpr or hp.max_relative_position <= 31)
            and (not core.aan or os.environ.get("ZERO_HIP_DECODE_FUSE_LN", "1") != "0") and not hp.use_ffn)


def _fuse_tail():
    import os
    return True


def _transposed(core, name):
    """bf16 copy of a projection weight with the input dimension contiguous (row = output channel): the operand layout of
    the fused decode kernels' matrix-core fragments.  Made by zk_transpose_bf16 once per WEIGHT VERSION (an optimiser
    update, a checkpoint load or an EMA swap makes a new one: variables.VariableStore.weight_version), not per
    encoded batch."""
    W = core.store.s(name)
    cache = core.__dict__.setdefault("_wt_cache", {})
    ver = core.store.weight_version
    ent = cache.get(name)
    if ent is None or ent[0] != ver or ent[2] != core.eng.realloc_gen:
        buf = core.eng.buf("dc.wt." + name, (W.shape[1], W.shape[0]), W.dtype)
        core.eng.lib.call("zk_transpose_bf16", W.data_ptr(), W.shape[1], buf.data_ptr(), W.shape[0], W.shape[0],
                          W.shape[1], core.eng.stream)
        ent = cache[name] = (ver, Mat(buf, W.shape[1], W.shape[0]), core.eng.realloc_gen)
    return ent[1]


def _cross_unfused(core, e, hp, state, lay, x, p, pre, l, time, time_dev):
    """Encoder-decoder attention sub-layer of a decode step, one launch per op."""
    H, nh, d = core.H, core.nh, core.d
    BK, K, Ls = state["BK"], state["K"], state["Ls"]
    q = e.mat("dc.q", BK, H)
    core._linear(x, p + "q_map", q)
    att = e.mat("dc.att", BK, H)
    rk = core.store.s(p + "rpr_keys/embeddings") if core.rpr else None
    rv = core.store.s(p + "rpr_values/embeddings") if core.rpr else None
    e.attn_fwd(q, lay["mk"], lay["mv"], att, None, BK, nh, 1, Ls, d, kmask=state["mask"], causal=False,
               q_pos0=time if time is not None else 0, rpr_k=rk, rpr_v=rv, max_rel=hp.max_relative_position, bsq=H,
               bsk=Ls * 2 * H, bsv=Ls * 2 * H, kv_group=K, pos_dev=time_dev, pos_flags=1)
    if core.fuse:
        # func.py:258-272: v_q = v_map(query); cache += v_q; o += cache / (time + 1)
        vq = e.mat("dc.vq", BK, H)
        core._linear(x, p + "v_map", vq)
        e.lib.call("zk_fuse_decode", vq.ptr, lay["aan"].data_ptr(), att.ptr, BK, H,
                   1.0 if time_dev is not None else 1.0 / float(time + 1),
                   time_dev.data_ptr() if time_dev is not None else None, e.stream)
    y = e.mat("dc.y", BK, H)
    core._linear(att, p + "o_map", y)
    x = core._ln_fwd(x, y, pre + "/" + core.cross, "dc%d.ca" % l, False, 0.0, 0)
    return x


def make_infer_fns(params, model_name):
    hp = params

    def encoding_fn(source, beam_size=None, max_steps=None):
        core = get_core(hp, model_name)
        e, H = core.eng, core.H
        K = hp.beam_size if beam_size is None else beam_size
        import os
        pad = max(1, int(os.environ.get("ZERO_HIP_DECODE_PAD_LEN", "8")))
        if _f32.wanted(hp):
            # the fp32 mode (round 5): fp32 masters, activations and caches through zk_f32_* (models/_decode_f32.py)
            from zero_amd.models._core import trim_columns
            return finish_state(_f32.encoding_state(core, hp, source, K, max_steps, DecodeStateF32, pad, trim_columns))
        if pad > 1:
            # Shape bucketing for the step-graph cache: the source is padded (id 0 = pad: masked in the encoder's
            # self-attention and in every cross-attention, func.py:372-387) and the cache length rounded up to a multiple
            # of `pad`, so that length-sorted batches fall into few shapes.  Masked keys contribute exact zeros to the
            # softmax sums: hypotheses and scores are unchanged (tests/test_gpu_model.py).
            from zero_amd.models._core import trim_columns
            src_np = trim_columns(np.asarray(source.cpu() if torch.is_tensor(source) else source))
            src_np = np.pad(src_np, ((0, 0), (0, -src_np.shape[1] % pad)))
            batch = core.upload(src_np, trim=False)
            if max_steps is not None:
                max_steps = -(-int(max_steps) // pad) * pad
        else:
            batch = core.upload(source)
        B, Ls = batch["B"], batch["Ls"]
        enc, smask = core.encode(batch, False, False)
        enc_keep = e.mat("dc.enc", B * Ls, H)
        enc_keep.t.copy_(enc.t)
        mask_keep = e.buf("dc.smask", (B, Ls), F32)
        mask_keep.copy_(smask)
        if max_steps is None:
            src_len = (np.asarray(source.cpu() if torch.is_tensor(source) else source) != 0).sum(1)
            max_steps = -(-(int(src_len.max()) + hp.decode_length + 2) // pad) * pad
        BK = B * K
        state = DecodeState()
        state.update({"_core": core, "B": B, "K": K, "BK": BK, "Ls": Ls, "Tmax": max_steps,
                      "encodes": enc_keep, "mask": mask_keep, "time_filled": 0,
                      "decoder": {"state": {}}})
        for l in range(hp.num_decoder_layer):
            p = "decoder/layer_%d/%s/dot_attention/" % (l, core.cross)
            kv = e.mat("dc%d.kv" % l, B * Ls, 2 * H)
            core._linear(enc_keep, p + "k_map", kv.cols_slice(0, H))
            core._linear(enc_keep, p + "v_map", kv.cols_slice(H, 2 * H))
            lay = {"mk": kv.cols_slice(0, H), "mv": kv.cols_slice(H, 2 * H)}
            if core.aan or core.fuse:
                lay["aan"] = None
            else:
                lay["k"] = lay["v"] = None
            state["decoder"]["state"]["layer_%d" % l] = lay
        state["_pp"] = 0
        nl = hp.num_decoder_layer
        state["wt"] = {}
        # (the fused launches keep a sentence's scores in LDS: 64 bytes per key for 16 rows)
        # and the whole workgroup state must fit the 160 KiB of a CU: otherwise 'wt' stays empty and the step takes the
        # launch-per-op path (_cross_unfused) instead of failing mid-decode
        if _fuse_att_ok(core, hp, K) and max(Ls, max_steps) <= 1024 and \
                e.lib.query("zk_dec_attn_lds", H, max(Ls, max_steps),
                            hp.max_relative_position if core.rpr else -1) <= 160 * 1024:
            for l in range(nl):
                blocks = [(core.cross, ("q_map", "o_map"))]
                if not core.aan:
                    blocks.append(("self_attention", ("qkv_map", "o_map")))
                for blk, maps in blocks:
                    for m in maps:
                        nm = "decoder/layer_%d/%s/dot_attention/%s/W_0_0" % (l, blk, m)
                        state["wt"][nm] = _transposed(core, nm)
        if core.aan or core.fuse:
            e.zero(e.buf("dc.aan.0", (nl, BK, H), F32))
            e.buf("dc.aan.1", (nl, BK, H), F32)
        else:
            for nm in ("k", "v"):
                for half in (0, 1):
                    e.buf("dc.%s.%d" % (nm, half), (nl, BK, max_steps, H))
        state.bind_caches()
        state["zero_flag"] = e.buf("dc.zflag", (1,), torch.int32)
        return finish_state(state)

    def finish_state(state):
        """The part of a batch's state that does not depend on the step's dtype: the packed host <-> device buffers of
        the search and the graph table."""
        core = state["_core"]
        e = core.eng
        B, K, BK = state["B"], state["K"], state["BK"]
        # per-step scalars live in device memory ({time, float bits of the length penalty, EOS-ban id}):
        # a captured decode-step graph reads the current values at replay time
        # what the host hands to a (replayed) step -- last tokens, previous log-probs, beam reorder index and
        # the per-step scalars {time, float bits of the length penalty, EOS-ban id} -- is ONE pinned buffer
        # and one async copy; what comes back (top-2K scores and flat indices) is one copy as well
        pack = e.buf("bs.pack", (3 * BK + 4,), torch.int32)
        state["pack_dev"] = pack
        pins = core.__dict__.setdefault("_decode_pins", {})       # pinned staging survives the batch (pinning ~0.2 ms)

        def pinned(name, n):
            t = pins.get(name)
            if t is None or t.numel() < n:
                t = pins[name] = torch.zeros(n + n // 4, dtype=torch.int32).pin_memory()
            t[:n].zero_()
            return t[:n]
        state["pack_host"] = pinned("pack", 3 * BK + 4)
        state["tok"] = pack[0:BK]
        state["prev"] = pack[BK:2 * BK].view(torch.float32)
        state["idx"] = pack[2 * BK:3 * BK]
        state["stepbuf"] = pack[3 * BK:3 * BK + 4]
        out = e.buf("bs.out", (2, B, 2 * K), torch.int32)
        state["out_dev"] = out
        state["out_host"] = pinned("out", 2 * B * 2 * K).view(2, B, 2 * K)
        state["ts"] = out[0].view(torch.float32)
        state["ti"] = out[1]
        state["graphs"] = {}
        # every launch argument of a step is static: the time step (cache slot, number of valid keys,
        # relative-position origin) is read from device memory by the kernels
        state["static_ok"] = True
        return state

    def step_static(state, temperature, forbid_value):
        _step_static(state, temperature, forbid_value)
        if state.get("_startup_held") and _startup_settled(state):
            startup_end(state)

    def _step_static(state, temperature, forbid_value):
        """One whole decode step with every per-step value read from device memory: beam reorder of
        the caches (indices chosen by the previous step), the AAN decoder step, logits, fused
        log-softmax + length penalty + top-2K.  The launch sequence is identical every step (per
        ping-pong parity), so after one eager pass per parity it is captured into a hipGraph."""
        core = state["_core"]
        e = core.eng
        parity = state["_pp"]
        book = state.get("book")          # device-resident search bookkeeping (search._beam_search_device)
        if "_gkey" not in state:
            adopt_graphs(state, book, temperature, forbid_value, hp.enable_noise_beam_search)
        g = state["graphs"].get(parity)

        def body():
            sb = state["stepbuf"]
            if book is not None:
                e.lib.call("zk_beam_dev_prepare", *book, e.stream)
            import os
            state.reorder(state["idx"], time_dev=sb[0:1],
                          defer_aan=core.aan)
            logits, _ = _step_cache(state["tok"], state, None, time_dev=sb[0:1])
            if hp.enable_noise_beam_search:      # search.py:143-145; a fresh stream position every step
                e.lib.call("zk_add_gumbel", logits.ptr, state["BK"], core.V, logits.ld, float(zdtype.epsilon()),
                           e.seed.data_ptr(), 7001, e.stream)
                e.lib.call("zk_seed_advance", e.seed.data_ptr(), 1, e.stream)
            fused_tail = False
            if book is not None and _fuse_tail():
                # merge of the chunked top-2K and the alive / finished bookkeeping in one launch
                ws = e.workspace(e.lib.query("zk_beam_topk_workspace", state["B"], state["K"], 2 * state["K"]))
                e.lib.ncalls += 1
                rc = e.lib.raw("zk_beam_topk_advance")(logits.ptr, logits.ld, float(temperature), float(forbid_value),
                                                       ws.data_ptr(), ws.numel(), *book, e.stream)
                if rc not in (0, -2):
                    e.lib.call("zk_beam_topk_advance", logits.ptr, logits.ld, float(temperature), float(forbid_value),
                               ws.data_ptr(), ws.numel(), *book, e.stream)      # raises with the library's message
                fused_tail = rc == 0
            if not fused_tail:
                e.beam_topk(logits, state["prev"], state["ts"], state["ti"], state["B"], state["K"], core.V,
                            2 * state["K"], temperature, 1.0, -1, forbid_value, scal_dev=sb[1:3])
                if book is not None:
                    e.lib.call("zk_beam_dev_advance", *book, e.stream)
        if g is None and core.__dict__.get("_decode_warm_rows", 0) >= state["BK"]:
            # A batch of at least this many beam rows has already been decoded on this engine, so the step's
            # scratch buffers exist: capture straight away instead of spending an eager pass first.  If the
            # capture does hit an allocation after all (a larger source length can grow a workspace), undo
            # the python-side ping-pong flip of the aborted pass and take the eager route below.
            pp0, gen0 = state["_pp"], e.realloc_gen
            try:
                gexec = e.graph_capture(body)
                core._decode_step_launches = e.last_graph_nodes
                if e.realloc_gen != gen0:
                    e.lib.call("zk_graph_destroy", gexec)
                    raise RuntimeError("buffer replaced during capture")
                state["graphs"][parity] = gexec
                e.graph_launch(gexec)
                return
            except Exception:
                torch.cuda.synchronize(e.device)
                state["_pp"] = pp0
                state.bind_caches()
                core._decode_warm_rows = 0
        if g is None:
            state["graphs"][parity] = "warm"
            body()
        elif g == "warm":
            # capture: python-side ping-pong bookkeeping runs during capture exactly as in an eager call.  A capture
            # that fails (an allocation met it: a workspace grew, or another thread allocated -- ROCm 7 invalidates
            # captures across threads) must not cost the evaluation, and the training run with it: undo the
            # python-side flip and run this step eagerly; the next step of this parity tries again.
            pp0 = state["_pp"]
            try:
                gexec = e.graph_capture(body)
            except Exception:
                torch.cuda.synchronize(e.device)
                state["_pp"] = pp0
                state.bind_caches()
                body()
                return
            state["graphs"][parity] = gexec
            core._decode_step_launches = e.last_graph_nodes
            e.graph_launch(gexec)
        else:
            state["_pp"] = 1 - state["_pp"]           # replay: redo the python-side pointer flip
            state.bind_caches()
            e.graph_launch(g)

    def _step_cache(target, state, time, time_dev=None):
        core = state["_core"]
        e, H, nh, d = core.eng, core.H, core.nh, core.d
        BK, K, B, Ls, Tmax = state["BK"], state["K"], state["B"], state["Ls"], state["Tmax"]
        if time_dev is None and time >= Tmax:
            raise RuntimeError("decode step %d exceeds the allocated cache length %d" % (time, Tmax))
        if state.get("f32"):
            return _f32.step_cache(target, state, time, time_dev, hp)
        import os as _os
        zf = state["zero_flag"]
        fuse_head = True
        if not fuse_head:
            e.lib.call("zk_all_equal", target.data_ptr(), BK, hp.tgt_vocab.pad(), zf.data_ptr(), e.stream)
        import os as _os
        fuse_ln = core.aan and _os.environ.get("ZERO_HIP_DECODE_FUSE_LN", "1") != "0"
        # the feed-forward pair as one launch: measured slower (EXPERIMENTS=1 library only, profiles/r04_negative_results.txt)
        FFN_PAIR = _os.environ.get("ZERO_HIP_DECODE_FFN_PAIR", "0") == "1" and e.lib.experiments
        # an attention sub-layer (projection, attention, the head's share of the output projection) as ONE launch per
        # (sentence, head) with the previous LayerNorm as its prologue (zk_dec_cross / zk_dec_self)
        fuse_att = bool(state.get("wt")) and _fuse_att_ok(core, hp, K)
        eps = zdtype.epsilon()
        tdev = time_dev.data_ptr() if time_dev is not None else None

        def ln_args(pro):
            """(x, ybuf, gamma, beta, xout, H, eps, z, cat_in, parts, nparts, part_stride, bias, cache, cat_out,
            inv_count, time_dev) of zk_dec_* / zk_ln_decode's row-local LayerNorm; pro None: no prologue."""
            if pro is None:
                return None
            g = lambda k: pro.get(k)
            return (g("x"), g("ybuf"), g("gamma"), g("beta"), g("out"), H, eps, g("z"), g("cat"), g("parts"),
                    g("nparts") or 0, g("stride") or 0, g("bias"), None, None, 1.0, None)

        def ln_scope(scope):
            return dict(gamma=core.b(scope + "/layer_norm/scale").data_ptr(),
                        beta=core.b(scope + "/layer_norm/offset").data_ptr())

        def rpr_tabs(p):
            """(rpr_k, rpr_v, max_rel) of the attention scope p: relative positions inside the fused launch (round 4)."""
            if not core.rpr:
                return (None, None, 0)
            return (core.store.s(p + "rpr_keys/embeddings").data_ptr(), core.store.s(p + "rpr_values/embeddings").data_ptr(),
                    hp.max_relative_position)

        def dec_cross(x_in, pro, p, lay, parts):
            la = ln_args(pro) or (x_in.ptr, None, None, None, None, H, eps, None, None, None, 0, 0, None, None, None,
                                  1.0, None)
            Wq, Wo = state["wt"][p + "q_map/W_0_0"], state["wt"][p + "o_map/W_0_0"]
            e.lib.call("zk_dec_cross", *la, Wq.ptr, Wq.ld, core.b(p + "q_map/b_0").data_ptr(), lay["mk"].ptr,
                       lay["mv"].ptr, lay["mk"].ld, lay["mv"].ld, Ls * 2 * H, Ls * 2 * H, state["mask"].data_ptr(), Ls,
                       Wo.ptr, Wo.ld, parts.data_ptr(), B, K, nh, Ls, float(d) ** -0.5, zdtype.inf(),
                       *(rpr_tabs(p) + (0 if time_dev is not None else time, tdev)), e.stream)

        def dec_self(x_in, pro, p, lay, parts):
            la = ln_args(pro) or (x_in.ptr, None, None, None, None, H, eps, None, None, None, 0, 0, None, None, None,
                                  1.0, None)
            Wq, Wo = state["wt"][p + "qkv_map/W_0_0"], state["wt"][p + "o_map/W_0_0"]
            e.lib.call("zk_dec_self", *la, Wq.ptr, Wq.ld, core.b(p + "qkv_map/b_0").data_ptr(), lay["k"].data_ptr(),
                       lay["v"].data_ptr(), Tmax, 0 if time_dev is not None else time, tdev, Wo.ptr, Wo.ld,
                       parts.data_ptr(), B, K, nh, float(d) ** -0.5, *rpr_tabs(p), e.stream)

        def ln_parts(x_in, scope, p, parts, tag):
            """x = LayerNorm(x_in + bf16(sum of the nh partial output projections + o_map bias))"""
            out = e.mat(tag + ".o", BK, H)
            ls = ln_scope(scope)
            e.lib.call("zk_ln_decode", x_in.ptr, e.mat("dc.y", BK, H).ptr, ls["gamma"], ls["beta"], out.ptr, BK, H, eps,
                       None, None, parts.data_ptr(), nh, BK * H, core.b(p + "o_map/b_0").data_ptr(), None, None, 1.0,
                       None, e.stream)
            return out
        ffn_split = 4 if fuse_att else 0

        def ffn_parts(x_in, f, l):
            """feed-forward sub-layer up to the output projection, left as split-K partial products (zk_gemm_parts: 64
            workgroups instead of 16 on 128 rows); the LayerNorm that follows adds them and the bias.  Returns the
            arguments of the row-local LayerNorm form (ln_args / zk_ln_decode)."""
            import ctypes
            hh = e.mat("dc%d.ff.h" % l, BK, core.F)
            W2 = core.W(f + "/ffn_layer/output/W_0_0")
            parts = e.buf("dc.ff.parts.%d" % (l & 1), (ffn_split, BK, H), F32)
            npair = None
            if FFN_PAIR and e.gemm_impl == 0:
                # both products in one launch, a barrier among its 64 workgroups between them (zk_ffn_pair)
                npair = e.ffn_pair(x_in, core.W(f + "/ffn_layer/enlarge/W_0_0"), core.b(f + "/ffn_layer/enlarge/b_0"), hh, W2,
                                   parts, ffn_split)
            if npair is not None:
                return dict(x=x_in.ptr, parts=parts.data_ptr(), nparts=npair, stride=BK * H,
                            bias=core.b(f + "/ffn_layer/output/b_0").data_ptr(), **ln_scope(f))
            core._linear(x_in, f + "/ffn_layer/enlarge", hh, act=1)
            n = ctypes.c_int(0)
            e.lib.call("zk_gemm_parts", hh.ptr, W2.ptr, parts.data_ptr(), BK, H, core.F, hh.ld, W2.ld, 0, 0, ffn_split,
                       ctypes.byref(n), e.stream)
            return dict(x=x_in.ptr, parts=parts.data_ptr(), nparts=n.value, stride=BK * H,
                        bias=core.b(f + "/ffn_layer/output/b_0").data_ptr(), **ln_scope(f))
        pend = None          # base model: the feed-forward LayerNorm of the previous layer, left to the next prologue
        x = e.mat("dc.x", BK, H)
        if fuse_head:
            # all-pad test + embedding + timing (+ the first layer's average-attention update) in one launch
            lay0 = state["decoder"]["state"]["layer_0"]
            aan0 = core.aan
            gat = state.pop("_aan_gather", None)       # the beam reorder of the running sums, deferred to this launch
            e.lib.call("zk_dec_embed", target.data_ptr(), hp.tgt_vocab.pad(), core.store.s(core.tgt_emb).data_ptr(),
                       core.b("bias").data_ptr(), e.timing(Tmax + 1, H).data_ptr(), x.ptr, BK, H, float(H) ** 0.5,
                       0 if time_dev is not None else time, time_dev.data_ptr() if time_dev is not None else None,
                       lay0["aan"].data_ptr() if aan0 else None, e.mat("dc.cat", BK, 2 * H).ptr if aan0 else None,
                       1.0 if time_dev is not None else 1.0 / float(time + 1),
                       gat[0].data_ptr() if gat else None, gat[1].data_ptr() if gat else None,
                       hp.num_decoder_layer if gat else 0, e.stream)
        else:
            e.embed_fwd(target, core.store.s(core.tgt_emb), core.b("bias"), x, BK, 1, H,
                        pos0=0 if time_dev is not None else time, zero_flag=zf, pos0_dev=time_dev, max_pos=Tmax)
        for l in range(hp.num_decoder_layer):
            pre = "decoder/layer_%d" % l
            lay = state["decoder"]["state"]["layer_%d" % l]
            if core.aan:
                a = pre + "/average_attention"
                cat = e.mat("dc.cat", BK, 2 * H)
                inv = 1.0 if time_dev is not None else 1.0 / float(time + 1)
                tdev = time_dev.data_ptr() if time_dev is not None else None
                if not (fuse_ln and l > 0 and not hp.use_ffn) and not (fuse_head and l == 0):
                    # else the previous layer's last LayerNorm (or the input launch) did it
                    e.lib.call("zk_aan_decode", x.ptr, lay["aan"].data_ptr(), cat.ptr, BK, H, inv, tdev, e.stream)
                if hp.use_ffn:           # transformer_aan.py:176-183
                    ya = e.mat("dc.ya", BK, H)
                    e.lib.call("zk_gather_rows", cat.ptr + H * 2, 2 * H * 2, None, ya.ptr, H * 2, BK, H * 2, e.stream)
                    hh = e.mat("dc.aah", BK, core.F)
                    core._linear(ya, a + "/ffn_layer/enlarge", hh, act=1)
                    core._linear(hh, a + "/ffn_layer/output", cat.cols_slice(H, 2 * H))
                z = e.mat("dc.z", BK, 2 * H)
                gate_split = 1 if fuse_att else 0
                if gate_split <= 1:
                    core._linear(cat, a + "/z_project", z)
                g = e.mat("dc.y", BK, H)
                if fuse_att:
                    # gate + residual + LayerNorm ride as the prologue of the encoder-decoder attention launch
                    xo = e.mat("dc%d.aa.o" % l, BK, H)
                    pend = dict(x=x.ptr, ybuf=g.ptr, out=xo.ptr, z=z.ptr, cat=cat.ptr, **ln_scope(a))
                    if gate_split > 1:
                        # z_project (K = 2H) as split-K partial products, summed (+ bias) by the prologue
                        import ctypes
                        Wz = core.W(a + "/z_project/W_0_0")
                        zp = e.buf("dc.z.parts", (gate_split, BK, 2 * H), F32)
                        n = ctypes.c_int(0)
                        e.lib.call("zk_gemm_parts", cat.ptr, Wz.ptr, zp.data_ptr(), BK, 2 * H, 2 * H, cat.ld, Wz.ld, 0, 0,
                                   gate_split, ctypes.byref(n), e.stream)
                        pend.update(z=None, parts=zp.data_ptr(), nparts=n.value, stride=BK * 2 * H,
                                    bias=core.b(a + "/z_project/b_0").data_ptr())
                    x = xo
                elif fuse_ln:
                    # gate + residual + LayerNorm in one launch (zk_ln_decode)
                    xo = e.mat("dc%d.aa.o" % l, BK, H)
                    e.lib.call("zk_ln_decode", x.ptr, g.ptr, core.b(a + "/layer_norm/scale").data_ptr(),
                               core.b(a + "/layer_norm/offset").data_ptr(), xo.ptr, BK, H, zdtype.epsilon(),
                               z.ptr, cat.ptr, None, 0, 0, None, None, None, 1.0, None, e.stream)
                    x = xo
                else:
                    e.aan_gate_fwd(z, cat, g, BK, H)
                    x = core._ln_fwd(x, g, a, "dc%d.aa" % l, False, 0.0, 0)
            elif fuse_att:
                p = pre + "/self_attention/dot_attention/"
                parts_s = e.buf("dc.parts.s", (nh, BK, H), F32)
                dec_self(x, pend, p, lay, parts_s)
                if pend is not None:
                    x = pend["xmat"]
                xo = e.mat("dc%d.sa.o" % l, BK, H)
                pend = dict(x=x.ptr, ybuf=e.mat("dc.y", BK, H).ptr, out=xo.ptr, parts=parts_s.data_ptr(), nparts=nh,
                            stride=BK * H, bias=core.b(p + "o_map/b_0").data_ptr(),
                            **ln_scope(pre + "/self_attention"))
                x = xo
            elif not core.fuse:
                p = pre + "/self_attention/dot_attention/"
                qkv = e.mat("dc.qkv", BK, 3 * H)
                core._linear(x, p + "qkv_map", qkv)
                for nm, c0 in (("k", H), ("v", 2 * H)):
                    if time_dev is not None:
                        e.lib.call("zk_cache_rows", qkv.ptr + c0 * 2, 3 * H * 2, None, lay[nm].data_ptr(),
                                   Tmax * H * 2, BK, H * 2, Tmax, time_dev.data_ptr(), 0, 0, e.stream)
                    else:
                        e.lib.call("zk_gather_rows", qkv.ptr + c0 * 2, 3 * H * 2, None,
                                   lay[nm].data_ptr() + time * H * 2, Tmax * H * 2, BK, H * 2, e.stream)
                att = e.mat("dc.att", BK, H)
                rk = core.store.s(p + "rpr_keys/embeddings") if core.rpr else None
                rv = core.store.s(p + "rpr_values/embeddings") if core.rpr else None
                e.attn_fwd(qkv.cols_slice(0, H), Mat(lay["k"], BK * Tmax, H), Mat(lay["v"], BK * Tmax, H), att,
                           None, BK, nh, 1, Tmax if time_dev is not None else time + 1, d, kmask=None, causal=False,
                           q_pos0=0 if time_dev is not None else time, rpr_k=rk,
                           rpr_v=rv, max_rel=hp.max_relative_position, bsq=3 * H, bsk=Tmax * H, bsv=Tmax * H,
                           pos_dev=time_dev, pos_flags=3)
                y = e.mat("dc.y", BK, H)
                core._linear(att, p + "o_map", y)
                x = core._ln_fwd(x, y, pre + "/self_attention", "dc%d.sa" % l, False, 0.0, 0)
            p = pre + "/" + core.cross + "/dot_attention/"
            if fuse_att:
                parts_c = e.buf("dc.parts.c", (nh, BK, H), F32)
                dec_cross(x, pend, p, lay, parts_c)        # x is already the prologue's output buffer
                pend = None
                x = ln_parts(x, pre + "/" + core.cross, p, parts_c, "dc%d.ca" % l)
            else:
                x = _cross_unfused(core, e, hp, state, lay, x, p, pre, l, time, time_dev)
            nxt = state["decoder"]["state"].get("layer_%d" % (l + 1))
            if fuse_att and not core.aan:
                # feed-forward sub-layer; its residual + LayerNorm is the prologue of the next layer's self-attention
                f = pre + "/feed_forward"
                xo = e.mat("dc%d.ff.o" % l, BK, H)
                if ffn_split > 1:
                    la = ffn_parts(x, f, l)
                else:
                    hh = e.mat("dc%d.ff.h" % l, BK, core.F)
                    core._linear(x, f + "/ffn_layer/enlarge", hh, act=1)
                    y = e.mat("dc%d.ff.y" % (l & 1), BK, H)
                    core._linear(hh, f + "/ffn_layer/output", y)
                    la = dict(x=x.ptr, ybuf=y.ptr, **ln_scope(f))
                la.update(out=xo.ptr, xmat=xo)
                if nxt is not None:
                    pend = la
                else:
                    e.lib.call("zk_ln_decode", *ln_args(la)[:5], BK, *ln_args(la)[5:], e.stream)
                    x = xo
                continue
            if fuse_ln and core.aan and not hp.use_ffn and (nxt is not None or ffn_split > 1):
                # feed-forward sub-layer whose LayerNorm also prepares the next layer's average attention
                # (cache += x; cat = [x | cache / (time + 1)]) in the same launch
                f = pre + "/feed_forward"
                xo = e.mat("dc%d.ff.o" % l, BK, H)
                if ffn_split > 1:
                    la = ffn_parts(x, f, l)
                else:
                    hh = e.mat("dc%d.ff.h" % l, BK, core.F)
                    core._linear(x, f + "/ffn_layer/enlarge", hh, act=1)
                    y = e.mat("dc.y", BK, H)
                    core._linear(hh, f + "/ffn_layer/output", y)
                    la = dict(x=x.ptr, ybuf=y.ptr, **ln_scope(f))
                g = lambda k: la.get(k)
                e.lib.call("zk_ln_decode", la["x"], g("ybuf"), la["gamma"], la["beta"], xo.ptr, BK, H, eps, None, None,
                           g("parts"), g("nparts") or 0, g("stride") or 0, g("bias"),
                           nxt["aan"].data_ptr() if nxt is not None else None,
                           e.mat("dc.cat", BK, 2 * H).ptr if nxt is not None else None,
                           1.0 if time_dev is not None else 1.0 / float(time + 1),
                           (time_dev.data_ptr() if time_dev is not None else None) if nxt is not None else None, e.stream)
                x = xo
            else:
                x = core._ffn_fwd(x, pre + "/feed_forward", "dc%d.ff" % l, False, 0, False)
        logits = e.mat("dc.logits", BK, core.Vpad, F32)
        e.gemm(x, core.W(core.soft_emb), logits, BK, core.V, H, 0, 1)
        if time_dev is None:
            state["time_filled"] = time + 1
        return logits, state

    def _step_dev(target, source, time):
        """transformer.py:277-281: encoder + training-path decoder on the whole prefix."""
        core = get_core(hp, model_name)
        e = core.eng
        batch = core.upload(source, target)
        enc, smask = core.encode(batch, False, False)
        feat, _, _ = core.decode_train(batch, enc, smask, False, False)
        BK, Lt = batch["B"], batch["Lt"]
        last = Mat(feat.t, BK, core.H, Lt * core.H, (Lt - 1) * core.H)
        logits = e.mat("dc.logits", BK, core.Vpad, F32)
        e.gemm(last, core.W(core.soft_emb), logits, BK, core.V, core.H, 0, 1)
        return logits, source

    def decoding_fn(target, state, time):
        if hp.search_mode == "cache":
            return _step_cache(target, state, time)
        if _f32.wanted(hp):
            from zero_amd.hip import ZeroHipError
            raise ZeroHipError("decode_dtype=float32 decodes with search_mode=cache only (the dev mode re-runs the bf16 "
                               "training-path decoder, transformer.py:277-281)")
        return _step_dev(target, state, time)

    decoding_fn.step_static = step_static
    return encoding_fn, decoding_fn
