# coding: utf-8
"""The fp32 decode mode (round 5; ``decode_dtype="float32"``): encoder pass + cached decoder step on fp32 master
weights, fp32 activations and fp32 accumulation -- the reference's own default dtype (utils/dtype.py:12-15; run.py
``default_dtype="float32"``).

Why it exists: the north star asks for token-id exact greedy decode against the fp32 reference.  The bf16 product path
cannot deliver that (round 4: the fp32 and the bf16-storage ORACLES disagree with each other on 3 % of 256 sentences);
a path that rounds where the reference rounds can.  It is the same search (zero_amd/search.py: device-resident
bookkeeping, step graphs, fused top-2K on fp32 logits) over a different step: one launch per op of
models/transformer.py:15-84 (encoder), 120-196 (decoder step) and transformer_aan.py:92-117, 165-192 on the kernels of
zero_amd/csrc/zk_f32.hip.  All four registered models: ``transformer``, ``transformer_aan`` (incl. ``use_ffn``),
``transformer_rpr`` (relative positions inside zk_f32_attn: modules/rpr.py:10-75) and ``transformer_fuse`` (merged attention,
func.py:258-275).

The state nest, caches and reorder follow models/_decode.py (beam-invariant tensors stored once per sentence, per-beam
caches double-buffered and reordered by one row gather); buffers are named ``dq.*`` so that a bf16 and an fp32 decode of
the same engine never alias.
"""

import numpy as np
import torch

from zero_amd.func import Mat
from zero_amd.utils import dtype as zdtype

F32 = torch.float32


def wanted(hp):
    return str(getattr(hp, "decode_dtype", "bfloat16")).lower() in ("float32", "fp32", "f32")


def check_model(core):
    if core.d % 4 != 0:
        from zero_amd.hip import ZeroHipError
        raise ZeroHipError("decode_dtype=float32 needs a head size that is a multiple of 4 (got %d)" % core.d)


class _Ops(object):
    """The op layer of the fp32 step: thin wrappers over the zk_f32_* entry points (include/zero_hip.h)."""

    def __init__(self, core):
        self.core, self.e = core, core.eng
        self.lib = core.eng.lib
        self.H, self.nh, self.d = core.H, core.nh, core.d
        import os
        # round 6: row-local launches folded into their neighbours (ZERO_HIP_F32_FUSE=0: one launch per op, as in round 5 --
        # the two forms are compared in tests/test_gpu_decode_f32.py)
        self.fold = os.environ.get("ZERO_HIP_F32_FUSE", "1") != "0"

    def mat(self, name, rows, cols):
        return Mat(self.e.buf("dq." + name, (rows, cols), F32), rows, cols)

    def w(self, name):
        t = self.core.store.w(name)              # fp32 master (physical shape)
        return Mat(t, t.shape[0], t.shape[1])

    def gemm(self, A, B, C, M, N, K, tb=0, bias=None, act=0):
        self.lib.call("zk_f32_gemm", A.ptr, B.ptr, C.ptr, M, N, K, A.ld, B.ld, C.ld, tb,
                      bias.data_ptr() if bias is not None else None, act, self.e.stream)

    def linear(self, x, scope, out, act=0):
        """func.py:14-65: out = x W + b (act 1: ReLU)."""
        W = self.w(scope + "/W_0_0")
        self.gemm(x, W, out, x.rows, W.cols, W.rows, 0, self.core.store.w(scope + "/b_0"), act)
        return out

    def add_ln(self, x, y, scope, out):
        """func.py:321-324 + 289-303: LN(x + y) with the scope's scale / offset."""
        st = self.core.store
        if self.fold and self.H % 4 == 0 and self.H <= 2048:
            # the 16-byte-load form (round 6: 4.8 us against 7.2 for a decode step's 128 rows)
            return self.ln_fused(x, y, scope, out)
        self.lib.call("zk_f32_add_ln", x.ptr, y.ptr if y is not None else None, st.w(scope + "/layer_norm/scale").data_ptr(),
                      st.w(scope + "/layer_norm/offset").data_ptr(), out.ptr, x.rows, self.H, zdtype.epsilon(), self.e.stream)
        return out

    def attn(self, q, k, v, out, B, Lq, Lk, bsq, bsk, bsv, kmask=None, ldmask=0, kv_group=1, nkeys_dev=None, rpr=None,
             q_pos0=0, q_pos_dev=None):
        """rpr: attention scope prefix (".../dot_attention/") whose rpr_keys / rpr_values tables take part, or None."""
        rk = rv = None
        if rpr is not None:
            rk = self.core.store.w(rpr + "rpr_keys/embeddings").data_ptr()
            rv = self.core.store.w(rpr + "rpr_values/embeddings").data_ptr()
        self.lib.call("zk_f32_attn", q.ptr, k.ptr, v.ptr, out.ptr, B, self.nh, Lq, Lk, self.d, q.ld, k.ld, v.ld, out.ld,
                      int(bsq), int(bsk), int(bsv), int(Lq * out.ld), kmask.data_ptr() if kmask is not None else None,
                      int(ldmask), int(kv_group), float(self.d) ** -0.5, zdtype.inf(),
                      nkeys_dev.data_ptr() if nkeys_dev is not None else None, rk, rv, int(self.core.hp.max_relative_position),
                      int(q_pos0), q_pos_dev.data_ptr() if q_pos_dev is not None else None, self.e.stream)
        return out

    def ln_fused(self, x, y, scope, out, gate=None, aan_next=None, time=0, time_dev=None):
        """zk_f32_ln_fused (round 6): LN(x + y) with row-local neighbours in the same launch.  gate = (z, cat): y is the
        average-attention gate of transformer_aan.py:186-189 and the residual is cat[:, :H] (x, y = None);
        aan_next = (cache tensor, cat Mat): the next layer's average attention from the normalised row."""
        st = self.core.store
        z, cat = gate if gate is not None else (None, None)
        cache, cat_out = aan_next if aan_next is not None else (None, None)
        self.lib.call("zk_f32_ln_fused", x.ptr if x is not None else None, y.ptr if y is not None else None,
                      z.ptr if z is not None else None, cat.ptr if cat is not None else None,
                      st.w(scope + "/layer_norm/scale").data_ptr(), st.w(scope + "/layer_norm/offset").data_ptr(), out.ptr,
                      out.rows, self.H, zdtype.epsilon(), cache.data_ptr() if cache is not None else None,
                      cat_out.ptr if cat_out is not None else None, int(time), time_dev, self.e.stream)
        return out

    def ffn(self, x, scope, tag, aan_next=None, time=0, time_dev=None):
        """func.py:327-338 + residual + LayerNorm (transformer.py:62-69)."""
        h = self.mat(tag + ".h", x.rows, self.core.F)
        self.linear(x, scope + "/ffn_layer/enlarge", h, act=1)
        y = self.mat("y", x.rows, self.H)
        self.linear(h, scope + "/ffn_layer/output", y)
        if aan_next is not None:
            return self.ln_fused(x, y, scope, self.mat(tag + ".o", x.rows, self.H), aan_next=aan_next, time=time,
                                 time_dev=time_dev)
        return self.add_ln(x, y, scope, self.mat(tag + ".o", x.rows, self.H))


def encode(core, hp, batch):
    """transformer.py:15-84 in fp32: -> (encoder output Mat [B*Ls, H], source mask fp32 [B, Ls])."""
    o = _Ops(core)
    e, H = core.eng, core.H
    B, Ls = batch["B"], batch["Ls"]
    T = B * Ls
    smask = batch["smask"]
    x = o.mat("enc.x0", T, H)
    e.lib.call("zk_f32_embed", batch["src"].data_ptr(), T, Ls, core.store.w(core.src_emb).data_ptr(),
               core.store.w("bias").data_ptr(), e.timing(Ls + 1, H).data_ptr(), int(e.timing(Ls + 1, H).shape[0]), x.ptr, H,
               float(H) ** 0.5, 0, None, None, e.stream)
    for l in range(hp.num_encoder_layer):
        pre = "encoder/layer_%d" % l
        p = pre + "/self_attention/dot_attention/"
        qkv = o.mat("enc.qkv", T, 3 * H)
        o.linear(x, p + "qkv_map", qkv)
        att = o.mat("enc.att", T, H)
        o.attn(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att, B, Ls, Ls,
               Ls * 3 * H, Ls * 3 * H, Ls * 3 * H, kmask=smask, ldmask=Ls, rpr=p if core.rpr else None)
        y = o.mat("y", T, H)
        o.linear(att, p + "o_map", y)
        x = o.add_ln(x, y, pre + "/self_attention", o.mat("e%d.sa.o" % l, T, H))
        x = o.ffn(x, pre + "/feed_forward", "e%d.ff" % l)
    return x, smask


def make_state_class(base):
    class DecodeStateF32(base):
        """models/_decode.DecodeState with 4-byte cache elements and ``dq.*`` buffers."""

        def reorder(self, index_dev, time_dev=None, defer_aan=False):
            core = self["_core"]
            e = core.eng
            BK, H, t = self["BK"], core.H, self["time_filled"]
            nl = core.hp.num_decoder_layer
            pp = self["_pp"]
            lay0 = self["decoder"]["state"]["layer_0"]
            if "aan" in lay0:
                src = e.buf("dq.aan.%d" % pp, (nl, BK, H), F32)
                dst = e.buf("dq.aan.%d" % (1 - pp), (nl, BK, H), F32)
                e.lib.call("zk_gather_rows_ex", src.data_ptr(), H * 4, index_dev.data_ptr(), dst.data_ptr(), H * 4,
                           nl * BK, H * 4, BK, e.stream)
            if "k" in lay0:
                Tmax = self["Tmax"]
                for nm in ("k", "v"):
                    src = e.buf("dq.%s.%d" % (nm, pp), (nl, BK, Tmax, H), F32)
                    dst = e.buf("dq.%s.%d" % (nm, 1 - pp), (nl, BK, Tmax, H), F32)
                    if time_dev is not None:
                        e.lib.call("zk_cache_rows", src.data_ptr(), Tmax * H * 4, index_dev.data_ptr(), dst.data_ptr(),
                                   Tmax * H * 4, nl * BK, H * 4, Tmax, time_dev.data_ptr(), 1, BK, e.stream)
                    else:
                        e.lib.call("zk_gather_rows_ex", src.data_ptr(), Tmax * H * 4, index_dev.data_ptr(),
                                   dst.data_ptr(), Tmax * H * 4, nl * BK, t * H * 4, BK, e.stream)
            self["_pp"] = 1 - pp
            self.bind_caches()

        def bind_caches(self):
            core = self["_core"]
            e = core.eng
            BK, H, nl, pp = self["BK"], core.H, core.hp.num_decoder_layer, self["_pp"]
            for l in range(nl):
                lay = self["decoder"]["state"]["layer_%d" % l]
                if "aan" in lay:
                    lay["aan"] = e.buf("dq.aan.%d" % pp, (nl, BK, H), F32)[l]
                if "k" in lay:
                    for nm in ("k", "v"):
                        lay[nm] = e.buf("dq.%s.%d" % (nm, pp), (nl, BK, self["Tmax"], H), F32)[l]
    return DecodeStateF32


def encoding_state(core, hp, source, K, max_steps, state_cls, pad, trim_columns):
    """encoding_fn of the fp32 mode: the state nest of models/_decode.py with fp32 tensors."""
    check_model(core)
    o = _Ops(core)
    e, H = core.eng, core.H
    if pad > 1:
        src_np = trim_columns(np.asarray(source.cpu() if torch.is_tensor(source) else source))
        src_np = np.pad(src_np, ((0, 0), (0, -src_np.shape[1] % pad)))
        batch = core.upload(src_np, trim=False)
        if max_steps is not None:
            max_steps = -(-int(max_steps) // pad) * pad
    else:
        batch = core.upload(source)
    B, Ls = batch["B"], batch["Ls"]
    enc, smask = encode(core, hp, batch)
    enc_keep = o.mat("enc", B * Ls, H)
    enc_keep.t.copy_(enc.t)
    mask_keep = e.buf("dq.smask", (B, Ls), F32)
    mask_keep.copy_(smask)
    if max_steps is None:
        src_len = (np.asarray(source.cpu() if torch.is_tensor(source) else source) != 0).sum(1)
        max_steps = -(-(int(src_len.max()) + hp.decode_length + 2) // pad) * pad
    BK = B * K
    state = state_cls()
    state.update({"_core": core, "B": B, "K": K, "BK": BK, "Ls": Ls, "Tmax": max_steps, "encodes": enc_keep,
                  "mask": mask_keep, "time_filled": 0, "decoder": {"state": {}}, "f32": True, "wt": {}})
    nl = hp.num_decoder_layer
    for l in range(nl):
        p = "decoder/layer_%d/%s/dot_attention/" % (l, core.cross)
        kv = o.mat("%d.kv" % l, B * Ls, 2 * H)
        o.linear(enc_keep, p + "k_map", kv.cols_slice(0, H))
        o.linear(enc_keep, p + "v_map", kv.cols_slice(H, 2 * H))
        lay = {"mk": kv.cols_slice(0, H), "mv": kv.cols_slice(H, 2 * H)}
        if core.aan or core.fuse:
            lay["aan"] = None
        else:
            lay["k"] = lay["v"] = None
        state["decoder"]["state"]["layer_%d" % l] = lay
    state["_pp"] = 0
    if core.aan or core.fuse:
        e.zero(e.buf("dq.aan.0", (nl, BK, H), F32))
        e.buf("dq.aan.1", (nl, BK, H), F32)
    else:
        for nm in ("k", "v"):
            for half in (0, 1):
                e.buf("dq.%s.%d" % (nm, half), (nl, BK, max_steps, H), F32)
    state.bind_caches()
    state["zero_flag"] = e.buf("dq.zflag", (1,), torch.int32)
    return state


def step_cache(target, state, time, time_dev, hp):
    """One cached decoder step in fp32 (transformer.py:88-196 with cache / transformer_aan.py:165-260): -> logits Mat
    fp32 [B*K, Vpad].  time_dev: the step counter lives in device memory (hipGraph replay)."""
    core = state["_core"]
    o = _Ops(core)
    e, H = core.eng, core.H
    BK, K, B, Ls, Tmax = state["BK"], state["K"], state["B"], state["Ls"], state["Tmax"]
    if time_dev is None and time >= Tmax:
        raise RuntimeError("decode step %d exceeds the allocated cache length %d" % (time, Tmax))
    tdev = time_dev.data_ptr() if time_dev is not None else None
    t_host = 0 if time_dev is not None else time
    fold = o.fold
    x = o.mat("x", BK, H)
    tim = e.timing(Tmax + 1, H)
    nl = hp.num_decoder_layer
    cat = o.mat("cat", BK, 2 * H) if core.aan else None
    aan_in_ln = fold and core.aan           # a layer's average attention is computed by the launch that produces its input
    if fold:
        lay0 = state["decoder"]["state"]["layer_0"]
        e.lib.call("zk_f32_embed_step", target.data_ptr(), BK, core.store.w(core.tgt_emb).data_ptr(),
                   core.store.w("bias").data_ptr(), tim.data_ptr(), int(tim.shape[0]), x.ptr, H, float(H) ** 0.5, t_host, tdev,
                   hp.tgt_vocab.pad(), lay0["aan"].data_ptr() if aan_in_ln else None, cat.ptr if aan_in_ln else None, e.stream)
    else:
        zf = state["zero_flag"]
        e.lib.call("zk_all_equal", target.data_ptr(), BK, hp.tgt_vocab.pad(), zf.data_ptr(), e.stream)
        e.lib.call("zk_f32_embed", target.data_ptr(), BK, 1, core.store.w(core.tgt_emb).data_ptr(), core.store.w("bias").data_ptr(),
                   tim.data_ptr(), int(tim.shape[0]), x.ptr, H, float(H) ** 0.5, t_host, tdev, zf.data_ptr(), e.stream)
    for l in range(nl):
        pre = "decoder/layer_%d" % l
        lay = state["decoder"]["state"]["layer_%d" % l]
        nxt = state["decoder"]["state"].get("layer_%d" % (l + 1))
        aan_next = (nxt["aan"], cat) if (aan_in_ln and nxt is not None) else None
        if core.aan:
            a = pre + "/average_attention"
            if not aan_in_ln:
                e.lib.call("zk_f32_aan_step", x.ptr, lay["aan"].data_ptr(), cat.ptr, BK, H, t_host, tdev, e.stream)
            if hp.use_ffn:           # transformer_aan.py:176-183: the averaged half goes through its own feed-forward
                hh = o.mat("aah", BK, core.F)
                o.linear(cat.cols_slice(H, 2 * H), a + "/ffn_layer/enlarge", hh, act=1)
                ya = o.mat("ya", BK, H)
                o.linear(hh, a + "/ffn_layer/output", ya)
                e.lib.call("zk_gather_rows", ya.ptr, H * 4, None, cat.ptr + H * 4, 2 * H * 4, BK, H * 4, e.stream)
            z = o.mat("z", BK, 2 * H)
            o.linear(cat, a + "/z_project", z)
            if fold:
                x = o.ln_fused(None, None, a, o.mat("d%d.aa.o" % l, BK, H), gate=(z, cat))
            else:
                g = o.mat("y", BK, H)
                e.lib.call("zk_f32_gate", z.ptr, cat.ptr, g.ptr, BK, H, e.stream)
                x = o.add_ln(x, g, a, o.mat("d%d.aa.o" % l, BK, H))
        elif not core.fuse:
            p = pre + "/self_attention/dot_attention/"
            qkv = o.mat("qkv", BK, 3 * H)
            o.linear(x, p + "qkv_map", qkv)
            for nm, c0 in (("k", H), ("v", 2 * H)):
                if time_dev is not None:
                    e.lib.call("zk_cache_rows", qkv.ptr + c0 * 4, 3 * H * 4, None, lay[nm].data_ptr(), Tmax * H * 4, BK,
                               H * 4, Tmax, tdev, 0, 0, e.stream)
                else:
                    e.lib.call("zk_gather_rows", qkv.ptr + c0 * 4, 3 * H * 4, None, lay[nm].data_ptr() + time * H * 4,
                               Tmax * H * 4, BK, H * 4, e.stream)
            att = o.mat("att", BK, H)
            kc, vc = Mat(lay["k"], BK * Tmax, H), Mat(lay["v"], BK * Tmax, H)
            # one query per beam row over the positions 0 .. time of ITS cache (no padding mask on the target side,
            # transformer.py:136; causality is the cache's length)
            o.attn(qkv.cols_slice(0, H), kc, vc, att, BK, 1, Tmax if time_dev is not None else time + 1, 3 * H, Tmax * H,
                   Tmax * H, nkeys_dev=time_dev, rpr=p if core.rpr else None, q_pos0=t_host, q_pos_dev=time_dev)
            y = o.mat("y", BK, H)
            o.linear(att, p + "o_map", y)
            x = o.add_ln(x, y, pre + "/self_attention", o.mat("d%d.sa.o" % l, BK, H))
        # encoder-decoder attention over the sentence's keys / values (stored once per sentence: kv_group = K)
        p = pre + "/" + core.cross + "/dot_attention/"
        qm = o.mat("q", BK, H)
        o.linear(x, p + "q_map", qm)
        att = o.mat("att", BK, H)
        # (relative positions, transformer_rpr.py:167-169: the query sits at position `time` against the SOURCE positions)
        o.attn(qm, lay["mk"], lay["mv"], att, BK, 1, Ls, H, Ls * 2 * H, Ls * 2 * H, kmask=state["mask"], ldmask=Ls, kv_group=K,
               rpr=p if core.rpr else None, q_pos0=t_host, q_pos_dev=time_dev)
        if core.fuse:
            # func.py:258-275: v_q = v_map(query); aan_o = (v_q + cache) / (time + 1); cache += v_q; o += aan_o
            vq = o.mat("vq", BK, H)
            o.linear(x, p + "v_map", vq)
            cat = o.mat("cat", BK, 2 * H)
            e.lib.call("zk_f32_aan_step", vq.ptr, lay["aan"].data_ptr(), cat.ptr, BK, H, t_host, tdev, e.stream)
            e.lib.call("zk_f32_add_rows", att.ptr, att.ld, cat.ptr + H * 4, 2 * H, att.ptr, att.ld, BK, H, e.stream)
        y = o.mat("y", BK, H)
        o.linear(att, p + "o_map", y)
        x = o.add_ln(x, y, pre + "/" + core.cross, o.mat("d%d.ca.o" % l, BK, H))
        x = o.ffn(x, pre + "/feed_forward", "d%d.ff" % l, aan_next=aan_next, time=t_host, time_dev=tdev)
    logits = Mat(e.buf("dq.logits", (BK, core.Vpad), F32), BK, core.Vpad)
    o.gemm(x, o.w(core.soft_emb), logits, BK, core.V, H, tb=1)
    if time_dev is None:
        state["time_filled"] = time + 1
    return logits, state
