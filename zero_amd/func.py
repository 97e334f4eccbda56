# coding: utf-8
"""Operator layer: the reference's func.py surface, each op one HIP launch.

Counterpart of the reference's func.py (linear :14-65, dot_attention :164-286,
layer_norm :289-303, residual_fn :321-324, ffn_layer :327-338,
add_timing_signal :341-369, attention_bias :372-400) -- but every function here
enqueues hand-written gfx950 kernels from libzero_hip.so on the current HIP
stream instead of building TF graph nodes.  Tensors are torch-ROCm storage
only: a ``Mat`` is (storage, rows, cols, leading dim, element offset) and what
crosses the C-ABI is its device address.

No op here has a CPU or torch fallback; if the library is missing the import of
``zero_amd.hip`` raises.
"""

import ctypes
import math
import os

import numpy as np
import torch

from zero_amd import hip
from zero_amd.utils import dtype as zdtype


class _GroupDesc(ctypes.Structure):
    """Mirror of ``struct GroupDesc`` (zero_amd/csrc/zk_gemm2.hip, include/zero_hip.h)."""
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ("M", "N", "K", "lda", "ldb", "ldc", "out_f32", "tile_start",
                                            "tiles_n", "ldr")] + [("colsum", ctypes.c_void_p), ("pad", ctypes.c_long)]


RED_COLS = 32      # columns per block of k_reduce_grouped (ZK_RED_COLS in zero_amd/csrc/zk_elem.hip)


class _FoldDesc(ctypes.Structure):
    """Mirror of ``struct FoldDesc`` (zero_amd/csrc/zk_prep.hip, include/zero_hip.h zk_ln_fold)."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("W", "gamma", "beta", "b", "Wf", "c", "d")] + \
               [(n, ctypes.c_int) for n in ("K", "N", "block_start", "pad")]


class _ColsumDesc(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("partials", ctypes.c_void_p)] + \
               [(n, ctypes.c_int) for n in ("rows", "N", "lda", "gy", "block_start", "pad")]


class _ReduceDesc(ctypes.Structure):
    _fields_ = [("partials", ctypes.c_void_p), ("out", ctypes.c_void_p * 3)] + \
               [(n, ctypes.c_int) for n in ("nblk", "nq", "H", "block_start", "bstride", "qstride")]


class Mat(object):
    """Row-major matrix view over a torch tensor."""
    __slots__ = ("t", "rows", "cols", "ld", "off")

    def __init__(self, t, rows, cols, ld=None, off=0):
        self.t, self.rows, self.cols = t, int(rows), int(cols)
        self.ld = int(cols if ld is None else ld)
        self.off = int(off)

    @property
    def ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()

    def cols_slice(self, c0, c1):
        return Mat(self.t, self.rows, c1 - c0, self.ld, self.off + c0)

    def torch(self):
        """Materialise as a torch tensor (tests / debugging only)."""
        flat = self.t.reshape(-1)
        return torch.as_strided(flat, (self.rows, self.cols), (self.ld, 1), self.off)


def _impl_from_env(var):
    v = os.environ.get(var, "auto").lower()
    return {"auto": 0, "naive": 1, "ref": 1, "mfma": 2}.get(v, 0)


def timing_table(length, channels, min_timescale=1.0, max_timescale=1.0e4):
    """func.py:341-369 as an fp32 [length, channels] table: sin in the first
    channels/2 columns, cos in the next channels/2 (not interleaved)."""
    nts = channels // 2
    inc = math.log(float(max_timescale) / float(min_timescale)) / (float(nts) - 1)
    inv = min_timescale * np.exp(np.arange(nts, dtype=np.float64) * -inc)
    st = np.arange(length, dtype=np.float64)[:, None] * inv[None, :]
    sig = np.zeros((length, channels), dtype=np.float64)
    sig[:, :nts] = np.sin(st)
    sig[:, nts:2 * nts] = np.cos(st)
    return sig.astype(np.float32)


class Engine(object):
    """Owns the per-device plumbing (stream, scratch, seeds) and exposes the ops."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip.ZeroHipError("the HIP hot path needs a GPU device (got %s); there is no CPU fallback"
                                   % self.device)
        self.lib = hip.lib()
        self.bufs = {}
        self._ws = {}
        self.gemm_impl = _impl_from_env("ZERO_HIP_GEMM")
        self.attn_impl = _impl_from_env("ZERO_HIP_ATTN")
        self.cu_count = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == "cuda" else 256
        self.seed = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.realloc_gen = 0
        self._timing = None
        # layer programs (run_program): one persistent launch per sentence-local chain instead of one launch per op
        # layer program (zk_layer.hip): an experiment, only present in a `make EXPERIMENTS=1` library
        self.programs_enabled = os.environ.get("ZERO_HIP_PROGRAM", "0") != "0" and self.lib.experiments
        # relative positions folded into the attention forward tile (zk_attn_dev.h attn_fwd_tile<.., RPR>)
        self.rpr_fold = True           # (an attribute, not a switch: the kernel tests flip it to reach the decomposed form)
        # folded backward: 72 KB of LDS (two workgroups per CU) or every tile resident (151 KB, the first form)
        self.rpr_bwd_resident = False  # (the kernel test of the 72-KB form compares it with this, the first, form)
        self._prog_host = None

    # ---- plumbing -----------------------------------------------------------
    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def buf(self, name, shape, dt=torch.bfloat16):
        """Named persistent buffer (static address -> graph-capturable)."""
        shape = tuple(int(s) for s in shape)
        b = self.bufs.get(name)
        if b is None or b.dtype != dt or b.numel() < int(np.prod(shape)):
            if b is not None:
                # a buffer some captured hipGraph may point at is being replaced: holders of graphs compare
                # this counter before replaying (zero_amd/main.py Trainer)
                self.realloc_gen += 1
            b = torch.empty(int(np.prod(shape)), dtype=dt, device=self.device)
            self.bufs[name] = b
        return b[:int(np.prod(shape))].view(*shape)

    def mat(self, name, rows, cols, dt=torch.bfloat16):
        return Mat(self.buf(name, (rows, cols), dt), rows, cols)

    def workspace(self, nbytes):
        """Scratch of the CURRENT stream (kernels on different streams may run concurrently)."""
        key = self.stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            if ws is not None:
                self.realloc_gen += 1          # captured graphs point at the old scratch (see buf())
            ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def timing(self, length, H):
        if self._timing is None or self._timing.shape[0] < length or self._timing.shape[1] != H:
            if self._timing is not None:
                self.realloc_gen += 1          # captured graphs read the old table
            n = max(256, int(length))
            self._timing = torch.from_numpy(timing_table(n, H)).to(self.device)
        return self._timing

    def set_seed(self, value):
        self.seed.fill_(int(value))

    def h2d(self, dst, arr):
        """Small host array -> device tensor `dst` (same element count) as an ASYNCHRONOUS copy on the current stream
        through a ring of pinned staging slots (zero_amd/utils/queuer.py PinnedRing): the training loop never blocks on
        an upload unless it runs more than a few steps ahead of the device."""
        if getattr(self, "_pins", None) is None:
            from zero_amd.utils.queuer import PinnedRing
            self._pins = PinnedRing(self.device)
        self._pins.put(dst, arr)

    @staticmethod
    def _copy_many_args(chunk):
        n = len(chunk)
        assert all(d.numel() * d.element_size() == s_.numel() * s_.element_size() for d, s_ in chunk)
        return ((ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in chunk]),
                (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in chunk]),
                (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in chunk]), n)

    def copy_many(self, pairs):
        """[(dst tensor, src tensor)] (same byte counts, multiples of 4): one launch (zk_copy_many) per 16 pairs."""
        pairs = [(d, s_) for d, s_ in pairs if d.numel()]
        for i in range(0, len(pairs), 16):
            self.lib.call("zk_copy_many", *(self._copy_many_args(pairs[i:i + 16]) + (self.stream,)))

    def copy_many_fits(self, pairs):
        """True when these pairs make exactly one zk_copy_many launch (what a graph with an in-graph commit must hold)."""
        n = len([1 for d, _ in pairs if d.numel()])
        return 1 <= n <= int(self.lib.raw("zk_copy_many_max")())

    def graph_set_copy_many(self, exec_, pairs):
        """Rewrite the one zk_copy_many launch inside the captured graph ``exec_`` to these (dst, src) pairs (at most 16 with
        elements; the same filter as copy_many).  False when the pairs do not fit one launch."""
        pairs = [(d, s_) for d, s_ in pairs if d.numel()]
        if not 1 <= len(pairs) <= 16:
            return False
        self.lib.call("zk_graph_set_copy_many", exec_, *self._copy_many_args(pairs))
        return True

    @property
    def upload_stream(self):
        """Side stream on which Trainer.step uploads and prepares the NEXT batch while the previous step runs."""
        if getattr(self, "_upload_stream", None) is None:
            self._upload_stream = torch.cuda.Stream(self.device)
        return self._upload_stream

    def batch_prep(self, batch):
        """zk_batch_prep on an uploaded batch (TransformerCore.upload): masks, loss weights, rows grouped by id."""
        B, Ls, Lt = batch["B"], batch["Ls"], batch.get("Lt", 0)
        tgt = batch.get("tgt")
        ss, ts = batch.get("src_sort"), batch.get("tgt_sort")
        nbytes = (self.lib.query("zk_batch_prep_workspace", B * Ls) if ss else 0) + \
            (self.lib.query("zk_batch_prep_workspace", B * Lt) if ts else 0)
        ws = self.buf("prep.scratch" + batch.get("suffix", ""), (nbytes,), torch.uint8) if nbytes else None
        p = lambda d, k: d[k].data_ptr() if d else None
        self.lib.call("zk_batch_prep", batch["src"].data_ptr(), hip.ptr(tgt), B, Ls, Lt,
                      p(ss, "rows"), p(ss, "seg"), p(ss, "uid"), p(ss, "n"),
                      p(ts, "rows"), p(ts, "seg"), p(ts, "uid"), p(ts, "n"),
                      hip.ptr(batch.get("smask")), hip.ptr(batch.get("tmask")), hip.ptr(batch.get("tw")),
                      float(batch.get("tw_scale", 1.0)), int(batch.get("max_id", 0)), hip.ptr(ws), nbytes, self.stream)

    def zero(self, t):
        self.lib.call("zk_zero", t.data_ptr(), t.numel() * t.element_size(), self.stream)

    # ---- func.py:14-65 linear and its backward mirrors -------------------------
    def gemm(self, A, B, C, M, N, K, ta, tb, alpha=1.0, bias=None, residual=None, act=0, aux=None,
             aux_scale=1.0, drop_p=0.0, sid=0, impl=None):
        out_f32 = 1 if C.t.dtype == torch.float32 else 0
        ws_bytes = self.lib.query("zk_gemm_workspace", M, N, K)
        ws = self.workspace(ws_bytes)
        self.lib.call(
            "zk_gemm", A.ptr, B.ptr, C.ptr, M, N, K, A.ld, B.ld, C.ld, ta, tb, out_f32, alpha,
            hip.ptr(bias), residual.ptr if residual is not None else None,
            residual.ld if residual is not None else 0, act,
            aux.ptr if aux is not None else None, aux.ld if aux is not None else 0, aux_scale,
            float(drop_p), self.seed.data_ptr(), sid,
            self.gemm_impl if impl is None else impl, ws.data_ptr(), ws.numel(), self.stream)

    # ---- residual + LayerNorm folded into GEMM epilogues (zk_gemm_ln / zk_ln_fold / zk_add_ln_bwd_lazy) ----------------
    # ---- residual + LayerNorm inside the producing GEMM launch (zk_gemm_add_ln)
    def sync_ln_state(self, rows, N):
        """(slots, meta) of the in-launch LayerNorm: the exchange slots of the row blocks' workgroups (zero-filled once,
        shared by all calls of this engine: they run one after the other on its stream) and meta = int32 [4]: the epoch
        word (ln_epoch_bump), the error word."""
        need = self.lib.query("zk_gemm_add_ln_workspace", int(rows), int(N))
        st = self.__dict__.get("_sync_ln")
        if st is None or st[0].numel() < need:
            if st is not None:
                self.realloc_gen += 1          # captured graphs point at the old slots (see buf())
            slots = torch.empty(max(int(need), 1 << 20), dtype=torch.uint8, device=self.device)
            self.zero(slots)
            meta = st[1] if st is not None else torch.empty(4, dtype=torch.int32, device=self.device)
            if st is None:
                self.zero(meta)
            st = self._sync_ln = (slots, meta)
        return st

    def ln_epoch_bump(self):
        """Once per forward pass (and whenever the sites of a pass run out): later zk_gemm_add_ln launches never take a
        slot written before this for one of theirs."""
        _, meta = self.sync_ln_state(1, 64)
        self.lib.call("zk_ln_epoch_bump", meta.data_ptr(), self.stream)
        self._sync_site = 0

    def gemm_add_ln(self, A, B, M, N, K, bias, residual, gamma, beta, y, s_out=None, mean=None, rstd=None, drop_p=0.0,
                    sid=0):
        """y = LN(residual + dropout(A @ B + bias)) in one launch; s_out / mean / rstd: what add_ln_bwd reads."""
        if self.__dict__.get("_sync_site", 255) >= 255:
            self.ln_epoch_bump()
        self._sync_site += 1
        slots, meta = self.sync_ln_state(M, N)
        self.lib.call("zk_gemm_add_ln", A.ptr, B.ptr, M, N, K, A.ld, B.ld, hip.ptr(bias), residual.ptr, residual.ld,
                      float(drop_p), self.seed.data_ptr(), sid, gamma.data_ptr(), beta.data_ptr(), zdtype.epsilon(),
                      s_out.ptr if s_out is not None else None, y.ptr, hip.ptr(mean), hip.ptr(rstd), slots.data_ptr(),
                      slots.numel(), meta.data_ptr(), self._sync_site, meta.data_ptr() + 4, self.stream)

    def attn_out_ln(self, q, k, v, att, lse, B, nh, Lq, Lk, d, kmask, causal, attn_drop_p, attn_sid, Wo, bias, residual,
                    gamma, beta, y, s_out=None, mean=None, rstd=None, drop_p=0.0, sid=0, kv_group=1, proj=None):
        """Attention forward + o_map + residual + LayerNorm in one launch (zk_attn_out_ln).  False (nothing launched)
        when the shape is not covered: the caller issues attn_fwd and gemm_add_ln.
        proj = (x Mat, Wp Mat, bp, pro): the projection in front of the attention runs in the same launch
        (zk_proj_attn_out_ln; pro = 3: merged qkv_map, q / k / v column slices of one matrix; pro = 1: q_map)."""
        if self.__dict__.get("_sync_site", 255) >= 255:
            self.ln_epoch_bump()
        M, N = B * Lq, nh * d
        slots, meta = self.sync_ln_state(M, N)
        fl = self._sync_flags_buf(B, nh)
        args = (q.ptr, k.ptr, v.ptr, att.ptr, hip.ptr(lse), B, nh, Lq, Lk, d, q.ld, k.ld, v.ld, att.ld, hip.ptr(kmask),
                1 if causal else 0, float(d) ** -0.5, zdtype.inf(), float(attn_drop_p), self.seed.data_ptr(), attn_sid, kv_group,
                Wo.ptr, Wo.ld, hip.ptr(bias), residual.ptr, residual.ld, float(drop_p), sid, gamma.data_ptr(), beta.data_ptr(),
                zdtype.epsilon(), s_out.ptr if s_out is not None else None, y.ptr, hip.ptr(mean), hip.ptr(rstd),
                slots.data_ptr(), slots.numel(), fl.data_ptr(), fl.numel(), meta.data_ptr(), self._sync_site + 1,
                meta.data_ptr() + 4, self.stream)
        name = "zk_attn_out_ln"
        if proj is not None:
            px, pW, pb, pro = proj
            args = (px.ptr, px.ld, pW.ptr, pW.ld, hip.ptr(pb), pW.rows, pro) + args
            name = "zk_proj_attn_out_ln"
        self.lib.ncalls += 1
        rc = self.lib.raw(name)(*args)
        if rc == 2:
            return False
        if rc != 0:
            self.lib.call(name, *args)      # raises with the library's message
        self._sync_site += 1
        return True

    def _sync_flags_buf(self, B, nh):
        need = self.lib.query("zk_attn_out_ln_flags", B, nh)
        fl = self.__dict__.get("_sync_flags")
        if fl is None or fl.numel() < need:
            if fl is not None:
                self.realloc_gen += 1
            fl = self._sync_flags = torch.empty(max(int(need), 1 << 16), dtype=torch.uint8, device=self.device)
            self.zero(fl)
        return fl

    def attn_bwd_ln(self, q, k, v, out, lse, dq, dk, dv, B, nh, Lq, Lk, d, kmask, causal, attn_drop_p, attn_sid, oproj,
                    dA, W, residual, s, mean, rstd, gamma, dsum, dy_out, partials, drop_p=0.0, sid=0):
        """Attention backward (o_map dgrad folded in: oproj = (dY Mat, W_o Mat)) + dx = dA @ W^T + residual + the LayerNorm
        backward below, one launch (zk_attn_bwd_ln).  False (nothing launched) when the shape is not covered."""
        if self.__dict__.get("_sync_site", 255) >= 255:
            self.ln_epoch_bump()
        M, N = B * Lq, nh * d
        slots, meta = self.sync_ln_state(M, N)
        fl = self._sync_flags_buf(B, nh)
        dy, Wo = oproj
        assert partials.numel() * 4 >= B * 3 * N * 4
        args = (q.ptr, k.ptr, v.ptr, out.ptr, lse.data_ptr(), dq.ptr, dk.ptr, dv.ptr, B, nh, Lq, Lk, d, q.ld, k.ld, v.ld, out.ld,
                dq.ld, dk.ld, dv.ld, hip.ptr(kmask), 1 if causal else 0, float(d) ** -0.5, zdtype.inf(), float(attn_drop_p),
                self.seed.data_ptr(), attn_sid, dy.ptr, dy.ld, Wo.ptr, Wo.ld, Wo.cols, dA.ptr, dA.ld, W.ptr, W.ld, dA.cols,
                residual.ptr if residual is not None else None, residual.ld if residual is not None else 0, s.ptr,
                mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), float(drop_p), sid, dsum.ptr,
                dy_out.ptr if dy_out is not None else None, partials.data_ptr(), slots.data_ptr(), slots.numel(),
                fl.data_ptr(), fl.numel(), meta.data_ptr(), self._sync_site + 1, meta.data_ptr() + 4, self.stream)
        self.lib.ncalls += 1
        rc = self.lib.raw("zk_attn_bwd_ln")(*args)
        if rc == 2:
            return False
        if rc != 0:
            self.lib.call("zk_attn_bwd_ln", *args)      # raises with the library's message
        self._sync_site += 1
        return True

    def gemm_ln_bwd(self, dY, W, M, N, K, residual, s, mean, rstd, gamma, dsum, dy_out, partials, drop_p=0.0, sid=0):
        """dgrad dY @ W^T (+ residual) and the backward of the residual + LayerNorm it feeds, in one launch."""
        if self.__dict__.get("_sync_site", 255) >= 255:
            self.ln_epoch_bump()
        self._sync_site += 1
        slots, meta = self.sync_ln_state(M, N)
        assert partials.numel() * 4 >= self.lib.query("zk_gemm_ln_bwd_partials", M, N)
        self.lib.call("zk_gemm_ln_bwd", dY.ptr, W.ptr, M, N, K, dY.ld, W.ld, residual.ptr if residual is not None else None,
                      residual.ld if residual is not None else 0, s.ptr, mean.data_ptr(), rstd.data_ptr(),
                      gamma.data_ptr(), float(drop_p), self.seed.data_ptr(), sid, dsum.ptr,
                      dy_out.ptr if dy_out is not None else None, partials.data_ptr(), slots.data_ptr(), slots.numel(),
                      meta.data_ptr(), self._sync_site, meta.data_ptr() + 4, self.stream)

    def ffn_pair(self, x, W1, b1, h, W2, parts, splits):
        """h = relu(x @ W1 + b1) and the split-K partial products of h @ W2 in ONE launch (zk_ffn_pair: the decode step's
        feed-forward pair).  Returns the number of parts, or None (nothing launched) when the shape is not covered."""
        import ctypes
        M, F, H = x.rows, W1.cols, W2.cols
        key = "ffpair.cnt.%d.%d" % ((M + 63) // 64, F)
        fresh = key not in self.bufs
        cnt = self.buf(key, (2,), torch.int64)
        if fresh:
            self.zero(cnt)
        _, meta = self.sync_ln_state(1, 64)
        n = ctypes.c_int(0)
        args = (x.ptr, W1.ptr, hip.ptr(b1), h.ptr, W2.ptr, parts.data_ptr(), M, F, H, W1.rows, x.ld, W1.ld, W2.ld, int(splits),
                ctypes.byref(n), cnt.data_ptr(), meta.data_ptr() + 4, self.stream)
        self.lib.ncalls += 1
        rc = self.lib.raw("zk_ffn_pair")(*args)
        if rc == 2:
            return None
        if rc != 0:
            self.lib.call("zk_ffn_pair", *args)      # raises with the library's message
        return n.value

    def sync_ln_usable(self):
        """One-time self-test of the in-launch exchange on this device (cached), over ALL THREE kernels that use it:
        zk_gemm_add_ln (a grid whose row blocks sit on one XCD each -- exchange through the L2 -- and one that straddles
        XCDs -- through memory), zk_gemm_ln_bwd (same two grids) and zk_attn_out_ln (sentence-aligned row tiles with
        Lq < 64: the shape whose 64-row tile reaches into the next sentence).  Each must finish without a workgroup giving
        up and satisfy a row property that needs every peer's partial: normalised rows have mean 0 / variance 1; the rows
        of a LayerNorm input gradient sum to 0 (sum_j rstd (g_j - mean g - xhat_j mean(g xhat)) = 0).  The exchange also
        assumes the part dispatches workgroup b to XCD b mod 8 (`sy_local`): a device that does not report 8 XCDs worth of
        CUs (another partition mode) is not trusted with it.  A platform that fails runs the two-launch structure --
        forward AND backward -- with a warning instead of wrong numbers."""
        ok = self.__dict__.get("_sync_ln_ok")
        if ok is not None:
            return ok
        ok = True
        dev = self.device
        try:
            props = torch.cuda.get_device_properties(dev)
            if props.multi_processor_count % 8 != 0 or props.multi_processor_count < 64:
                ok = False              # not the 8-XCD SPX layout the workgroup -> XCD rule was measured on
            g = torch.Generator(device="cpu").manual_seed(7)
            rnd = lambda *shape, scale=1.0: (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)

            def rows_normalised(y):
                yf = y.float()
                # (bf16 storage moves a row mean by ~3e-5; ONE missing peer partial moves it by ~1.6e-2)
                return bool(torch.isfinite(yf).all()) and float(yf.mean(1).abs().max()) <= 5e-3 and \
                    abs(float(yf.var(1, unbiased=False).mean()) - 1.0) <= 0.02
            for M in ((4096, 320) if ok else ()):
                N, K = 512, 64
                A, W, R = rnd(M, K), rnd(K, N, scale=0.1), rnd(M, N)
                y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                s_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
                ones, zeros = torch.ones(N, device=dev), torch.zeros(N, device=dev)
                self.ln_epoch_bump()
                self.gemm_add_ln(Mat(A, M, K), Mat(W, K, N), M, N, K, None, Mat(R, M, N), ones, zeros, Mat(y, M, N),
                                 s_out=Mat(s_out, M, N), mean=mean, rstd=rstd)
                torch.cuda.synchronize(dev)
                if self.sync_ln_errors() or not rows_normalised(y):
                    ok = False
                    break
                # backward: dx = dY W2^T + res feeds the backward of the LayerNorm whose (s, mean, rstd) the forward left
                dY, W2, res = rnd(M, K), rnd(N, K, scale=0.1), rnd(M, N, scale=0.1)
                ds = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                part = torch.empty(self.lib.query("zk_gemm_ln_bwd_partials", M, N) // 4, device=dev)
                self.gemm_ln_bwd(Mat(dY, M, K), Mat(W2, N, K), M, N, K, Mat(res, M, N), Mat(s_out, M, N), mean, rstd, ones,
                                 Mat(ds, M, N), None, part)
                torch.cuda.synchronize(dev)
                dsf = ds.float()
                # (bf16 storage leaves |row sum| / row L1 norm at ~3e-5; one missing peer partial: ~1.6e-2)
                if self.sync_ln_errors() or not bool(torch.isfinite(dsf).all()) or \
                        float((dsf.sum(1).abs() / dsf.abs().sum(1).clamp_min(1e-6)).max()) > 2e-3:
                    ok = False
                    break
            if ok:
                # attention + o_map + LayerNorm, Lq < 64 (sentence-aligned tiles), one XCD-local grid and one straddling
                for B in (64, 9):
                    nh, L, d = 8, 40, 64
                    H = nh * d
                    qkv, Wo, R = rnd(B * L, 3 * H, scale=0.5), rnd(H, H, scale=0.05), rnd(B * L, H)
                    att = torch.empty(B * L, H, dtype=torch.bfloat16, device=dev)
                    y = torch.empty(B * L, H, dtype=torch.bfloat16, device=dev)
                    lse = torch.empty(B * nh * L, device=dev)
                    ones, zeros = torch.ones(H, device=dev), torch.zeros(H, device=dev)
                    qm = Mat(qkv, B * L, 3 * H)
                    self.ln_epoch_bump()
                    done = self.attn_out_ln(qm.cols_slice(0, H), qm.cols_slice(H, 2 * H), qm.cols_slice(2 * H, 3 * H),
                                            Mat(att, B * L, H), lse, B, nh, L, L, d, None, False, 0.0, 0, Mat(Wo, H, H), None,
                                            Mat(R, B * L, H), ones, zeros, Mat(y, B * L, H))
                    torch.cuda.synchronize(dev)
                    if done and (self.sync_ln_errors() or not rows_normalised(y)):
                        ok = False
                        break
        except hip.ZeroHipError:
            ok = False
        if not ok:
            import logging
            logging.getLogger("zero_amd").warning(
                "the in-launch LayerNorm exchange failed its self-test on this device; running the two-launch structure "
                "(ZERO_HIP_SYNC_LN=0)")
            st = self.__dict__.get("_sync_ln")
            if st is not None:
                torch.cuda.synchronize(self.device)
                st[1][1:2].zero_()         # the error word of the failed probe must not fail a later check
                torch.cuda.synchronize(self.device)
        self._sync_ln_ok = ok
        return ok

    def sync_ln_err_ptr(self):
        """Device address of the exchange's give-up word (stable for the engine's lifetime): zk_adam_step's skip_word."""
        _, meta = self.sync_ln_state(1, 64)
        return meta.data_ptr() + 4

    def sync_ln_errors(self):
        """1 if a workgroup of some zk_gemm_add_ln launch ever gave up waiting for its peers (synchronises)."""
        st = self.__dict__.get("_sync_ln")
        if st is None:
            return 0
        torch.cuda.synchronize()
        return int(st[1][1].item())

    def gemm_ln(self, A, B, C, M, N, K, bias, np_, residual=None, act=0, drop_p=0.0, sid=0, stat_out=None, in_part=None,
                in_c=None, res_part=None, res_gamma=None, res_beta=None):
        """zk_gemm_ln: producer (stat_out) / lazy residual (res_part, res_gamma, res_beta) / consumer (in_part, in_c)."""
        self.lib.call("zk_gemm_ln", A.ptr, B.ptr, C.ptr, M, N, K, A.ld, B.ld, C.ld, hip.ptr(bias),
                      residual.ptr if residual is not None else None, residual.ld if residual is not None else 0, act,
                      float(drop_p), self.seed.data_ptr(), sid, hip.ptr(stat_out), hip.ptr(in_part), hip.ptr(in_c),
                      hip.ptr(res_part), hip.ptr(res_gamma), hip.ptr(res_beta), int(np_), zdtype.epsilon(), self.stream)

    def ln_fold(self, problems):
        """One launch: for every (W fp32 master [K, N], gamma [K], beta [K], bias [N] or None, Wf Mat bf16, c, d) write
        Wf = bf16(gamma o W), c = colsum(Wf), d = beta . W + bias.  The device descriptor table is cached."""
        key = tuple((w.data_ptr(), g.data_ptr(), wf.ptr) for w, g, _, _, wf, _, _ in problems)
        cache = self.__dict__.setdefault("_fold_cache", {})
        ent = cache.get(key)
        if ent is None:
            arr = (_FoldDesc * len(problems))()
            start = 0
            for i, (w, g, bt, b, wf, c, d) in enumerate(problems):
                K, N = w.shape
                assert N % 64 == 0 and wf.ld == N
                r = arr[i]
                r.W, r.gamma, r.beta, r.b = w.data_ptr(), g.data_ptr(), bt.data_ptr(), hip.ptr(b) or 0
                r.Wf, r.c, r.d = wf.ptr, c.data_ptr(), d.data_ptr()
                r.K, r.N, r.block_start, r.pad = K, N, start, 0
                start += N // 64
            ent = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(problems), start)
            cache[key] = ent
        dev, n, total = ent
        self.lib.call("zk_ln_fold", dev.data_ptr(), n, total, self.stream)

    def add_ln_bwd_lazy(self, dout, s, part, gamma, beta, y_out, dsum, dy, dgamma, dbeta, dbias_prev, drop_p=0.0, sid=0,
                        private_ws=None):
        ws_bytes = self.lib.query("zk_add_ln_bwd_workspace", dout.rows, dout.cols)
        ws = private_ws if private_ws is not None else self.workspace(ws_bytes)
        assert ws.numel() * ws.element_size() >= ws_bytes
        self.lib.call("zk_add_ln_bwd_lazy", dout.ptr, s.ptr, part.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                      y_out.ptr if y_out is not None else None, dsum.ptr, dy.ptr if dy is not None else None,
                      hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(dbias_prev), dout.rows, dout.cols, zdtype.epsilon(),
                      float(drop_p), self.seed.data_ptr(), sid, ws.data_ptr(), ws.numel() * ws.element_size(),
                      1 if private_ws is not None else 0, self.stream)

    def gemm_kseg(self, segments, C, M, N, kseg, tb, residual=None):
        """C = sum over (A_s, B_s) in segments of A_s @ B_s (B_s transposed when tb) in one launch
        (zk_gemm_kseg).  All A_s share a leading dimension, all B_s too."""
        n = len(segments)
        a = (ctypes.c_void_p * n)(*[s[0].ptr for s in segments])
        b = (ctypes.c_void_p * n)(*[s[1].ptr for s in segments])
        lda, ldb = segments[0][0].ld, segments[0][1].ld
        assert all(s[0].ld == lda and s[1].ld == ldb for s in segments)
        self.lib.call("zk_gemm_kseg", a, b, n, kseg, C.ptr, M, N, lda, ldb, C.ld, tb,
                      residual.ptr if residual is not None else None, residual.ld if residual is not None else 0,
                      self.stream)

    def gemm_grouped(self, problems, ta, tb, tile=128):
        """One launch for many independent GEMMs with the same ta/tb.
        problems: list of (A, B, C, M, N, K, bias-or-None[, residual Mat-or-None]) with Mat operands.
        The device descriptor table is cached per problem list (buffers are static, so it is built
        once)."""
        problems = [tuple(p) + (None,) * (9 - len(p)) for p in problems]     # ..., residual, column-sum output
        # 128, 64, (256, 128), (128, 256), (256, 256) [fp32 outputs only] or (256, 256, 0) = the same without spreading
        # the LDS-DMA issue between the MFMA groups
        k32 = isinstance(tile, tuple) and len(tile) == 3 and tile[2] == "k32"     # 256x256: four stages of 32-deep K tiles
        spread = not (isinstance(tile, tuple) and len(tile) == 3 and not tile[2]) and not k32
        bm, bn = (tile, tile) if isinstance(tile, int) else tile[:2]
        code = {(128, 128): 1, (64, 64): 4, (256, 128): 5, (128, 256): 6, (256, 256): 7 if spread else 8}[(bm, bn)]
        if (bm, bn) == (256, 256):
            # fp32 tile, + optional column sums of B when ta = 1, tb = 0: the bias gradient beside a weight gradient, by
            # two extra MFMAs per eight in the tm = 0 tiles
            assert all(bias is None and r is None and c.t.dtype == torch.float32 and (cs is None or (ta and not tb))
                       for _, _, c, _, _, _, bias, r, cs in problems)
            if any(p[8] is not None for p in problems):
                code |= 256
            if k32:
                assert ta and not tb, "the 32-deep ring exists for the weight-gradient form (ta = 1, tb = 0)"
                assert self.lib.experiments, "the 32-deep ring is an experiment: make EXPERIMENTS=1"
                code |= 512
        elif any(p[8] is not None for p in problems):
            assert code in (5, 6) and not tb, "column sums ride on the producer waves of the wide tiles (tb = 0)"
        key = (ta, tb, bm, bn, spread, k32) + tuple((a.ptr, b.ptr, c.ptr, M, N, K, hip.ptr(bias) or 0, r.ptr if r is not None else 0,
                                                hip.ptr(cs) or 0) for a, b, c, M, N, K, bias, r, cs in problems)
        cache = self.__dict__.setdefault("_group_cache", {})
        ent = cache.get(key)
        if ent is None:
            arr = (_GroupDesc * len(problems))()
            start = 0
            for i, (a, b, c, M, N, K, bias, res, cs) in enumerate(problems):
                tn = (N + bn - 1) // bn
                d = arr[i]
                d.A, d.B, d.C, d.bias = a.ptr, b.ptr, c.ptr, hip.ptr(bias) or 0
                d.res, d.ldr = (res.ptr, res.ld) if res is not None else (0, 0)
                d.colsum = hip.ptr(cs) or 0
                d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, a.ld, b.ld, c.ld
                d.out_f32 = 1 if c.t.dtype == torch.float32 else 0
                d.tile_start, d.tiles_n = start, tn
                start += ((M + bm - 1) // bm) * tn
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            ent = (host.to(self.device), len(problems), start)
            cache[key] = ent
        dev, n, total = ent
        self.lib.call("zk_gemm_grouped", dev.data_ptr(), n, total, ta, tb, code, self.stream)

    def gemm_grouped_update(self, problems, upd):
        """zk_gemm_grouped_update: every weight gradient of the step in one launch of 256 x 256 tiles (ta = 1, tb = 0, bias
        column sums riding along), and for the problems marked fusable (10th element of the tuple) the TF1 Adam update of
        the variable inside the same launch -- their gradient is never stored.  upd: dict(master, m, v, shadow, grad,
        hyper) of flat tensors.  Returns (ranges [(lo, hi)] of the flat buffers that were updated, sq tensor, n_extra)."""
        problems = [tuple(p) + (None,) * (10 - len(p)) for p in problems]
        grad = upd["grad"]
        g0, g1 = grad.data_ptr(), grad.data_ptr() + grad.numel() * 4
        key = ("upd",) + tuple((a.ptr, b.ptr, c.ptr, M, N, K, hip.ptr(cs) or 0, bool(fz))
                               for a, b, c, M, N, K, _, _, cs, fz in problems)
        cache = self.__dict__.setdefault("_group_cache", {})
        ent = cache.get(key)
        if ent is None:
            arr = (_GroupDesc * len(problems))()
            start, ranges = 0, []
            for i, (a, b, c, M, N, K, bias, res, cs, fz) in enumerate(problems):
                assert bias is None and res is None and c.t.dtype == torch.float32
                tn = (N + 255) // 256
                d = arr[i]
                d.A, d.B, d.C, d.bias, d.res, d.ldr = a.ptr, b.ptr, c.ptr, 0, 0, 0
                d.colsum = hip.ptr(cs) or 0
                d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, a.ld, b.ld, c.ld
                d.out_f32, d.tile_start, d.tiles_n = 1, start, tn
                fuse = bool(fz) and c.ld == N and g0 <= c.ptr and c.ptr + M * N * 4 <= g1
                d.pad = 1 if fuse else 0
                if fuse:
                    lo = (c.ptr - g0) // 4
                    ranges.append((lo, lo + M * N))
                start += ((M + 255) // 256) * tn
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            ent = (host.to(self.device), len(problems), start, tuple(sorted(ranges)))
            cache[key] = ent
        dev, n, total, ranges = ent
        sq = self.buf("upd.sq", (total * 16,), torch.float32)
        self.lib.call("zk_gemm_grouped_update", dev.data_ptr(), n, total, upd["master"].data_ptr(), upd["m"].data_ptr(),
                      upd["v"].data_ptr(), upd["shadow"].data_ptr(), g0, upd["hyper"].data_ptr(), sq.data_ptr(), self.stream)
        return ranges, sq, total * 8

    def reductions_grouped(self, colsums, ln_parts, rpr_parts=()):
        """colsums: [(Mat dY, out fp32 view, private fp32 partial buffer)];
        ln_parts: [(partials buffer, rows, H, dgamma, dbeta, dbias_prev-or-None[, True: written by gemm_ln_bwd])];
        rpr_parts: [(partials fp32 [slices][2][64*64] (device address), slices, n, d rpr_k, d rpr_v)] -- the table-gradient
        partials the folded relative-position backward left behind (attn_bwd(defer_tables=...)).
        Two launches: column partial sums of every dY, then every final reduction."""
        key = tuple((a.ptr, o.data_ptr()) for a, o, _ in colsums) + \
            tuple((lp[0].data_ptr(), lp[1], lp[6] if len(lp) > 6 else None, lp[3].data_ptr(), lp[4].data_ptr(), hip.ptr(lp[5])) for lp in ln_parts) + \
            tuple((pp, ns, n, dk.data_ptr()) for pp, ns, n, dk, _ in rpr_parts)
        cache = self.__dict__.setdefault("_red_cache", {})
        ent = cache.get(key)
        if ent is None:
            lib = self.lib
            cd = (_ColsumDesc * max(len(colsums), 1))()
            rd = (_ReduceDesc * max(len(colsums) + len(ln_parts) + len(rpr_parts), 1))()
            cstart = rstart = 0
            k = 0
            for i, (a, o, pw) in enumerate(colsums):
                gy = lib.raw("zk_colsum_rowchunks")(a.rows)
                assert pw.numel() >= gy * a.cols
                d = cd[i]
                d.a, d.partials, d.rows, d.N, d.lda, d.gy, d.block_start, d.pad = \
                    a.ptr, pw.data_ptr(), a.rows, a.cols, a.ld, gy, cstart, 0
                cstart += ((a.cols + 63) // 64) * gy
                r = rd[k]
                r.partials, r.nblk, r.nq, r.H, r.block_start = pw.data_ptr(), gy, 1, a.cols, rstart
                r.out[0], r.out[1], r.out[2] = o.data_ptr(), None, None
                rstart += (a.cols + RED_COLS - 1) // RED_COLS
                k += 1
            for lp in ln_parts:
                pw, rows, H, dg, db, dbp = lp[:6]
                # (7th element: the partials came from zk_gemm_ln_bwd -- one partial row per 64-row block)
                # (or an int: that many partial rows -- one per sentence from attn_bwd_ln)
                if len(lp) > 6:
                    nblk = (rows + 63) // 64 if lp[6] is True else int(lp[6])
                else:
                    nblk = lib.raw("zk_ln_bwd_blocks")(rows)
                r = rd[k]
                r.partials, r.nblk, r.nq, r.H, r.block_start = pw.data_ptr(), nblk, 3, H, rstart
                r.out[0], r.out[1], r.out[2] = dg.data_ptr(), db.data_ptr(), hip.ptr(dbp)
                rstart += 3 * ((H + RED_COLS - 1) // RED_COLS)
                k += 1
            for (pp, ns, n, dk, dv) in rpr_parts:
                r = rd[k]
                r.partials, r.nblk, r.nq, r.H, r.block_start = pp, ns, 2, n, rstart
                r.bstride, r.qstride = 2 * 64 * 64, 64 * 64
                r.out[0], r.out[1], r.out[2] = dk.data_ptr(), dv.data_ptr(), None
                rstart += 2 * ((n + RED_COLS - 1) // RED_COLS)
                k += 1
            cdev = torch.frombuffer(bytearray(bytes(cd)), dtype=torch.uint8).to(self.device)
            rdev = torch.frombuffer(bytearray(bytes(rd)), dtype=torch.uint8).to(self.device)
            ent = (cdev, len(colsums), cstart, rdev, k, rstart)
            cache[key] = ent
        cdev, nc, cblocks, rdev, nr, rblocks = ent
        if nc:
            self.lib.call("zk_colsum_grouped", cdev.data_ptr(), nc, cblocks, self.stream)
        if nr:
            self.lib.call("zk_reduce_grouped", rdev.data_ptr(), nr, rblocks, self.stream)

    def colsum(self, A, out, skip_L=0, accumulate=False, drop_p=0.0, sid=0):
        ws_bytes = self.lib.query("zk_colsum_workspace", A.rows, A.cols)
        ws = self.workspace(ws_bytes)
        self.lib.call("zk_colsum_ex", A.ptr, A.rows, A.cols, A.ld, out.data_ptr(), skip_L,
                      1 if accumulate else 0, float(drop_p), self.seed.data_ptr(), sid, ws.data_ptr(),
                      ws.numel(), self.stream)

    # ---- func.py:164-286 attention core ----------------------------------------
    # ---- attention (func.py:218-256; relative positions modules/rpr.py:10-75) -------------------
    def _rpr_tables(self, rpr_k, rpr_v, max_rel, d):
        """Zero-padded [nrp, d] bf16 copies of the two relative-position tables (nrp = 2*max_rel+1
        rounded up to 8) so that they can be GEMM operands with an 8-aligned contraction length."""
        nrel = 2 * max_rel + 1
        nrp = (nrel + 7) // 8 * 8
        if rpr_k.shape[0] >= nrp and rpr_v.shape[0] >= nrp:
            # views of the variable store: already padded with zero rows (zero_amd/variables.py)
            return nrp, Mat(rpr_k, nrp, d), Mat(rpr_v, nrp, d)
        pads = []
        for tag, t in (("k", rpr_k), ("v", rpr_v)):
            name = "rpr.pad%s.%d" % (tag, t.data_ptr())
            fresh = name not in self.bufs
            pad = self.mat(name, nrp, d)
            if fresh:
                self.zero(pad.t)           # buffers are uninitialised; the padding rows must stay zero
            self.lib.call("zk_gather_rows", t.data_ptr(), d * 2, None, pad.ptr, d * 2, nrel, d * 2, self.stream)
            pads.append(pad)
        return nrp, pads[0], pads[1]

    def _rpr_mfma(self, impl, d, Lk, rpr_k, lds, max_rel, bwd=False):
        """Can the relative-position attention run on the MFMA kernels (decomposed form)?"""
        impl = self.attn_impl if impl is None else impl
        return rpr_k is not None and impl in ((0, 2, 3) if bwd else (0, 2)) and d == 64 and Lk <= 256 and \
            2 * max_rel + 1 <= 64 and all(x % 8 == 0 for x in lds)

    def attn_fwd(self, q, k, v, out, lse, B, nh, Lq, Lk, d, kmask=None, causal=False, q_pos0=0,
                 rpr_k=None, rpr_v=None, max_rel=0, drop_p=0.0, sid=0, bsq=0, bsk=0, bsv=0, kv_group=1,
                 impl=None, pos_dev=None, pos_flags=0):
        gq = pb = None
        ldg = nrp = 0
        eff_impl = self.attn_impl if impl is None else impl
        fold = self.rpr_fold and rpr_k is not None and eff_impl in (0, 2) and d == 64 and 2 * max_rel + 1 <= 64
        if fold:
            # relative positions inside the MFMA tile (tables in LDS): no table products through HBM, no extra
            # launches; also serves single-query decode steps (falls back to the reference kernel when the shape is
            # not covered)
            eff_impl = (0 if eff_impl == 0 else 2) | 256
        elif Lq > 1 and kv_group == 1 and not bsq and self._rpr_mfma(impl, d, Lk, rpr_k, (q.ld, k.ld, v.ld, out.ld), max_rel):
            # decomposed form: scores gather Q_h.Rk^T, the kernel returns the per-index sums of P and
            # O += pb.Rv finishes the value term -- three grouped GEMM launches around the MFMA kernel
            T = B * Lq
            nrp, rk, rv = self._rpr_tables(rpr_k, rpr_v, max_rel, d)
            ldg = nh * nrp
            gq = self.mat("rpr.gq.%d" % T, T, ldg, torch.float32)
            pb = self.mat("rpr.pb.%d" % T, T, ldg)
            self.gemm_grouped([(q.cols_slice(h * d, (h + 1) * d), rk, gq.cols_slice(h * nrp, (h + 1) * nrp), T, nrp, d, None)
                               for h in range(nh)], 0, 1, tile=64)
        self.lib.call(
            "zk_attn_fwd", q.ptr, k.ptr, v.ptr, out.ptr, hip.ptr(lse), B, nh, Lq, Lk, d, q.ld, k.ld, v.ld,
            out.ld, hip.ptr(kmask), 1 if causal else 0, q_pos0, float(d) ** -0.5, zdtype.inf(),
            hip.ptr(rpr_k), hip.ptr(rpr_v), max_rel, float(drop_p), self.seed.data_ptr(), sid,
            bsq, bsk, bsv, kv_group, eff_impl, hip.ptr(pos_dev), pos_flags,
            gq.ptr if gq is not None else None, pb.ptr if pb is not None else None, ldg, nrp, self.stream)
        if gq is not None:
            self.gemm_grouped([(pb.cols_slice(h * nrp, (h + 1) * nrp), rv, out.cols_slice(h * d, (h + 1) * d), T, d, nrp,
                                None, out.cols_slice(h * d, (h + 1) * d)) for h in range(nh)], 0, 0, tile=64)

    def attn_bwd(self, q, k, v, out, dout, lse, dq, dk, dv, B, nh, Lq, Lk, d, kmask=None, causal=False,
                 rpr_k=None, rpr_v=None, drpr_k=None, drpr_v=None, max_rel=0, drop_p=0.0, sid=0, impl=None,
                 defer_tables=None, oproj=None):
        """oproj = (dY Mat, W_o Mat): `dout` is the product dY . W_o^T that has NOT been formed yet -- the single-tile
        kernel computes its 64 x 64 piece per (sentence, head) itself; when another kernel would run (return code 2,
        nothing launched) the product is formed here into `dout` and the call repeated without the pair."""
        if oproj is not None:
            dy, Wo = oproj
            rc = self._attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, nh, Lq, Lk, d, kmask, causal, rpr_k, rpr_v,
                                drpr_k, drpr_v, max_rel, drop_p, sid, impl, defer_tables,
                                (dy.ptr, dy.ld, Wo.ptr, Wo.ld, Wo.cols))
            if rc != 2:
                return
            self.gemm(dy, Wo, dout, dy.rows, Wo.rows, Wo.cols, 0, 1)
        self._attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, nh, Lq, Lk, d, kmask, causal, rpr_k, rpr_v, drpr_k,
                       drpr_v, max_rel, drop_p, sid, impl, defer_tables, (None, 0, None, 0, 0))

    def _attn_bwd(self, q, k, v, out, dout, lse, dq, dk, dv, B, nh, Lq, Lk, d, kmask, causal, rpr_k, rpr_v, drpr_k,
                  drpr_v, max_rel, drop_p, sid, impl, defer_tables, op):
        eff = self.attn_impl if impl is None else impl
        fold = self.rpr_fold and rpr_k is not None and drpr_k is not None and eff in (0, 2) and d == 64 and \
            Lq <= 64 and Lk <= 64 and 2 * max_rel + 1 <= 64 and rpr_k.shape[0] >= 2 * max_rel + 1
        if fold:
            # relative positions inside the single-tile backward kernel; table gradients are overwritten
            nbytes = self.lib.query("zk_attn_bwd_rpr_workspace", B, nh, Lq)
            if defer_tables is not None:
                # defer_tables = (list, tag): the per-(sentence, head) partials of the table gradients stay in a buffer
                # private to this attention and ONE grouped launch sums those of every layer (reductions_grouped)
                pend, tag = defer_tables
                ws = self.buf("g.%s.rprws" % tag, (nbytes,), torch.uint8)
            else:
                ws = self.workspace(nbytes)
            self.lib.ncalls += 1
            args = (q.ptr, k.ptr, v.ptr, out.ptr, dout.ptr, lse.data_ptr(), dq.ptr, dk.ptr, dv.ptr,
                    hip.ptr(drpr_k), hip.ptr(drpr_v), B, nh, Lq, Lk, d, q.ld, k.ld, v.ld, out.ld, dout.ld, dq.ld,
                    dk.ld, dv.ld, hip.ptr(kmask), 1 if causal else 0, 0, float(d) ** -0.5, zdtype.inf(),
                    hip.ptr(rpr_k), hip.ptr(rpr_v), max_rel, float(drop_p), self.seed.data_ptr(), sid,
                    (0 if eff == 0 else 2) | 256 | (512 if defer_tables is not None else 0) |
                    (1024 if self.rpr_bwd_resident else 0), ws.data_ptr(), ws.numel(),
                    None, None, None, None, 0, 0) + tuple(op) + (self.stream,)
            rc = self.lib.raw("zk_attn_bwd")(*args)
            if rc == 1:        # the folded kernel ran and left the partials: the caller's grouped reduction sums them
                defer_tables[0].append((ws.data_ptr() + B * nh * Lq * 4, B * nh, (2 * max_rel + 1) * d, drpr_k, drpr_v))
            elif rc == 2 and op[0] is not None:
                return 2
            elif rc != 0:
                self.lib.call("zk_attn_bwd", *args)      # raises with the library's message
            return 0
        ws_bytes = self.lib.query("zk_attn_bwd_workspace", B, nh, Lq)
        ws = self.workspace(ws_bytes)
        dec = self._rpr_mfma(impl, d, 0, rpr_k, (q.ld, k.ld, v.ld, out.ld, dout.ld, dq.ld, dk.ld, dv.ld), max_rel, bwd=True)
        if op[0] is not None:
            if dec or rpr_k is not None:
                return 2           # decomposed products / reference kernels read dout
            args = (q.ptr, k.ptr, v.ptr, out.ptr, dout.ptr, lse.data_ptr(), dq.ptr, dk.ptr, dv.ptr, None, None, B, nh, Lq,
                    Lk, d, q.ld, k.ld, v.ld, out.ld, dout.ld, dq.ld, dk.ld, dv.ld, hip.ptr(kmask), 1 if causal else 0, 0,
                    float(d) ** -0.5, zdtype.inf(), None, None, max_rel, float(drop_p), self.seed.data_ptr(), sid,
                    self.attn_impl if impl is None else impl, ws.data_ptr(), ws.numel(), None, None, None, None, 0, 0) + \
                tuple(op) + (self.stream,)
            self.lib.ncalls += 1
            rc = self.lib.raw("zk_attn_bwd")(*args)
            if rc not in (0, 2):
                self.lib.call("zk_attn_bwd", *args)      # raises with the library's message
            return rc
        tabs = (None, None, None, None)
        ldg = nrp = 0
        if not dec and drpr_k is not None:
            # the reference kernels ADD their table gradients with atomics
            self.zero(drpr_k)
            self.zero(drpr_v)
        if dec:
            T = B * Lq
            nrp, rk, rv = self._rpr_tables(rpr_k, rpr_v, max_rel, d)
            ldg = nh * nrp
            gq = self.mat("rpr.gq.%d" % T, T, ldg, torch.float32)
            gd = self.mat("rpr.gd.%d" % T, T, ldg, torch.float32)
            pb, dsb = self.mat("rpr.pb.%d" % T, T, ldg), self.mat("rpr.dsb.%d" % T, T, ldg)
            heads = range(nh)
            sl = lambda m, h, w: m.cols_slice(h * w, (h + 1) * w)
            self.gemm_grouped([(sl(q, h, d), rk, sl(gq, h, nrp), T, nrp, d, None) for h in heads] +
                              [(sl(dout, h, d), rv, sl(gd, h, nrp), T, nrp, d, None) for h in heads], 0, 1, tile=64)
            tabs = (gq.ptr, gd.ptr, pb.ptr, dsb.ptr)
        self.lib.call(
            "zk_attn_bwd", q.ptr, k.ptr, v.ptr, out.ptr, dout.ptr, lse.data_ptr(), dq.ptr, dk.ptr, dv.ptr,
            hip.ptr(drpr_k), hip.ptr(drpr_v), B, nh, Lq, Lk, d, q.ld, k.ld, v.ld, out.ld, dout.ld, dq.ld,
            dk.ld, dv.ld, hip.ptr(kmask), 1 if causal else 0, 0, float(d) ** -0.5, zdtype.inf(),
            hip.ptr(rpr_k), hip.ptr(rpr_v), max_rel, float(drop_p), self.seed.data_ptr(), sid,
            self.attn_impl if impl is None else impl, ws.data_ptr(), ws.numel(),
            tabs[0], tabs[1], tabs[2], tabs[3], ldg, nrp, None, 0, None, 0, 0, self.stream)
        if dec:
            # dQ += dsb.Rk ; table gradients: per-head partials dsb_h^T Q_h and pb_h^T dO_h, summed over heads
            self.gemm_grouped([(sl(dsb, h, nrp), rk, sl(dq, h, d), T, d, nrp, None, sl(dq, h, d)) for h in heads],
                              0, 0, tile=64)
            # table gradients dsb_h^T Q_h and pb_h^T dO_h: [nrp x d] outputs over K = T rows.  One 64x64 tile per head would
            # walk all T rows alone (64 K steps, 24 us on 16 workgroups); cutting the rows into KS slices (every (head,
            # slice) its own tile, zk_sum_slices adds the nh * KS partials) was measured and is SLOWER in the step
            # (same box: KS = 1 6.37 ms, 4 6.41, 8 6.64), so KS = 1 stays the default
            KS = 1
            Tk = T // KS
            part = self.mat("rpr.part", 2 * nh * KS * nrp, d, torch.float32)
            rows = lambda i: Mat(part.t, nrp, d, d, i * nrp * d)
            rsl = lambda m, h, w, s_: Mat(m.t, Tk, w, m.ld, m.off + s_ * Tk * m.ld + h * w)
            self.gemm_grouped([(rsl(dsb, h, nrp, s_), rsl(q, h, d, s_), rows(h * KS + s_), nrp, d, Tk, None)
                               for h in heads for s_ in range(KS)] +
                              [(rsl(pb, h, nrp, s_), rsl(dout, h, d, s_), rows((nh + h) * KS + s_), nrp, d, Tk, None)
                               for h in heads for s_ in range(KS)], 1, 0, tile=64)
            n = (2 * max_rel + 1) * d
            # every table belongs to ONE attention scope: its gradient is overwritten, not accumulated (no zero fill)
            self.lib.call("zk_sum_slices", drpr_k.data_ptr(), part.ptr, nh * KS, n, nrp * d, 0, self.stream)
            self.lib.call("zk_sum_slices", drpr_v.data_ptr(), part.ptr + nh * KS * nrp * d * 4, nh * KS, n, nrp * d, 0,
                          self.stream)

    # ---- embedding + timing (transformer.py:16-33, 88-119; func.py:341-369) -------
    def embed_fwd(self, ids, table, bias, out, B, L, H, shift=False, pos0=0, zero_flag=None, drop_p=0.0,
                  sid=0, pos0_dev=None, max_pos=None):
        tim = self.timing((max_pos if max_pos is not None else pos0) + L, H)
        self.lib.call("zk_embed_fwd", ids.data_ptr(), table.data_ptr(), bias.data_ptr(), tim.data_ptr(),
                      out.ptr, B, L, H, float(H) ** 0.5, 1 if shift else 0, pos0, hip.ptr(zero_flag),
                      float(drop_p), self.seed.data_ptr(), sid, hip.ptr(pos0_dev), self.stream)

    def embed_fwd_pair(self, ids_a, table_a, out_a, La, sid_a, ids_b, table_b, out_b, Lb, sid_b, bias, B, H, drop_p=0.0,
                       bump_epoch=False):
        """Encoder input (ids_a) and shifted decoder input (ids_b) of a training step in one launch (zk_embed_fwd_pair).
        bump_epoch: the launch also does what ln_epoch_bump() does (the forward pass starts with both)."""
        tim = self.timing(max(La, Lb), H)
        meta = self.sync_ln_state(1, 64)[1] if bump_epoch else None
        self.lib.call("zk_embed_fwd_pair", ids_a.data_ptr(), table_a.data_ptr(), out_a.ptr, La, sid_a, ids_b.data_ptr(),
                      table_b.data_ptr(), out_b.ptr, Lb, sid_b, bias.data_ptr(), tim.data_ptr(), B, H, float(H) ** 0.5,
                      float(drop_p), self.seed.data_ptr(), meta.data_ptr() if meta is not None else None, self.stream)
        if bump_epoch:
            self._sync_site = 0

    def beam_topk(self, logits, prev_lp, out_s, out_i, B, K, V, k2, temperature, penalty, forbid_id, forbid_value,
                  scal_dev=None):
        ws_bytes = self.lib.query("zk_beam_topk_workspace", B, K, k2)
        ws = self.workspace(ws_bytes)
        self.lib.call("zk_beam_topk", logits.ptr, prev_lp.data_ptr(), out_s.data_ptr(), out_i.data_ptr(), B, K, V,
                      logits.ld, k2, float(temperature), float(penalty), int(forbid_id), float(forbid_value),
                      hip.ptr(scal_dev), ws.data_ptr(), ws.numel(), self.stream)

    def embed_bwd(self, ids, dout, dtable, dbias, B, L, H, shift=False, drop_p=0.0, sid=0):
        self.lib.call("zk_embed_bwd", ids.data_ptr(), dout.ptr, dtable.data_ptr(), dbias.data_ptr(), B, L, H,
                      float(H) ** 0.5, 1 if shift else 0, float(drop_p), self.seed.data_ptr(), sid, self.stream)

    def embed_bwd_sorted(self, sort, dout, dtable, H, accumulate, drop_p=0.0, sid=0):
        """sort = dict(rows, seg, uid, n (device int), max_uniq) prepared by the host at upload time."""
        self.lib.call("zk_embed_bwd_sorted", sort["rows"].data_ptr(), sort["seg"].data_ptr(),
                      sort["uid"].data_ptr(), sort["n"].data_ptr(), sort["max_uniq"], dout.ptr,
                      dtable.data_ptr(), H, float(H) ** 0.5, 1 if accumulate else 0, float(drop_p),
                      self.seed.data_ptr(), sid, self.stream)

    def embed_bwd_sorted_pair(self, sort_a, dout_a, dtable_a, acc_a, sid_a, sort_b, dout_b, dtable_b, acc_b, sid_b, H, drop_p=0.0):
        """Both embedding tables' gradient scatters in one launch (different tables; zk_embed_bwd_sorted_pair)."""
        self.lib.call("zk_embed_bwd_sorted_pair", sort_a["rows"].data_ptr(), sort_a["seg"].data_ptr(), sort_a["uid"].data_ptr(),
                      sort_a["n"].data_ptr(), sort_a["max_uniq"], dout_a.ptr, dtable_a.data_ptr(), 1 if acc_a else 0, sid_a,
                      sort_b["rows"].data_ptr(), sort_b["seg"].data_ptr(), sort_b["uid"].data_ptr(), sort_b["n"].data_ptr(),
                      sort_b["max_uniq"], dout_b.ptr, dtable_b.data_ptr(), 1 if acc_b else 0, sid_b, H, float(H) ** 0.5,
                      float(drop_p), self.seed.data_ptr(), self.stream)

    def colsum_pair(self, A, skip_a, sid_a, B, skip_b, sid_b, out, drop_p=0.0):
        """out = colsum(A) + colsum(B) (rows r % skip == 0 left out when skip > 0): zk_colsum_pair."""
        assert A.cols == B.cols
        ws_bytes = self.lib.query("zk_colsum_workspace", A.rows, A.cols) + self.lib.query("zk_colsum_workspace", B.rows, B.cols)
        ws = self.workspace(ws_bytes)
        self.lib.call("zk_colsum_pair", A.ptr, A.rows, A.ld, skip_a, sid_a, B.ptr, B.rows, B.ld, skip_b, sid_b, A.cols,
                      out.data_ptr(), float(drop_p), self.seed.data_ptr(), ws.data_ptr(), ws.numel(), self.stream)

    # ---- residual + layer norm (func.py:289-303, 321-324) -------------------------
    def add_ln_fwd(self, x, y, gamma, beta, out, sum_out=None, mean=None, rstd=None, drop_p=0.0, sid=0):
        self.lib.call("zk_add_ln_fwd", x.ptr, y.ptr if y is not None else None, gamma.data_ptr(),
                      beta.data_ptr(), out.ptr, sum_out.ptr if sum_out is not None else None, hip.ptr(mean),
                      hip.ptr(rstd), x.rows, x.cols, zdtype.epsilon(), float(drop_p), self.seed.data_ptr(), sid,
                      self.stream)

    def add_ln_bwd(self, dout, s, mean, rstd, gamma, dsum, dy, dgamma, dbeta, dbias_prev, drop_p=0.0, sid=0,
                   private_ws=None):
        """private_ws: a buffer owned by this call -> the column reduction is deferred; finish it with
        :meth:`add_ln_bwd_reduce` (e.g. on the side stream)."""
        ws_bytes = self.lib.query("zk_add_ln_bwd_workspace", dout.rows, dout.cols)
        ws = private_ws if private_ws is not None else self.workspace(ws_bytes)
        assert ws.numel() * ws.element_size() >= ws_bytes
        self.lib.call("zk_add_ln_bwd", dout.ptr, s.ptr, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                      dsum.ptr, dy.ptr if dy is not None else None, hip.ptr(dgamma), hip.ptr(dbeta),
                      hip.ptr(dbias_prev), dout.rows, dout.cols, float(drop_p), self.seed.data_ptr(), sid,
                      ws.data_ptr(), ws.numel() * ws.element_size(), 1 if private_ws is not None else 0,
                      self.stream)

    def add_ln_bwd_reduce(self, ws, rows, H, dgamma, dbeta, dbias_prev):
        self.lib.call("zk_add_ln_bwd_reduce", ws.data_ptr(), rows, H, hip.ptr(dgamma), hip.ptr(dbeta),
                      hip.ptr(dbias_prev), self.stream)

    # ---- loss (util.py:88-103; transformer.py:198-216) ------------------------------
    def ce_fused(self, logits, ids, w, ce, dlogits, rows, V, label_smooth):
        self.lib.call("zk_ce_fused", logits.ptr, ids.data_ptr(), hip.ptr(w), hip.ptr(ce),
                      dlogits.ptr if dlogits is not None else None, rows, V, logits.ld, float(label_smooth),
                      self.stream)

    # ---- fused logits + cross entropy (training path; transformer.py:182-216) ------------------
    def logits_ce_fwd(self, feat, E, ids, ce, lse, T, V, label_smooth):
        ws_bytes = self.lib.query("zk_logits_ce_workspace", T, V)
        ws = self.workspace(ws_bytes)
        self.lib.call("zk_logits_ce_fwd", feat.ptr, E.ptr, ids.data_ptr(), hip.ptr(ce), lse.data_ptr(), T, V,
                      feat.cols, feat.ld, E.ld, float(label_smooth), ws.data_ptr(), ws.numel(), self.stream)

    def logits_ce_bwd(self, feat, E, ids, w, lse, dlogits, T, V, label_smooth):
        self.lib.call("zk_logits_ce_bwd", feat.ptr, E.ptr, ids.data_ptr(), w.data_ptr(), lse.data_ptr(),
                      dlogits.ptr, T, V, feat.cols, feat.ld, E.ld, dlogits.ld, float(label_smooth), self.stream)

    def target_stats(self, ids, mask, w, B, L, loss_scale=1.0):
        self.lib.call("zk_target_stats", ids.data_ptr(), hip.ptr(mask), hip.ptr(w), B, L, float(loss_scale),
                      self.stream)

    def loss_reduce(self, ce, ids, per_sample, loss, B, L):
        self.lib.call("zk_loss_reduce", ce.data_ptr(), ids.data_ptr(), hip.ptr(per_sample), hip.ptr(loss), B, L,
                      self.stream)

    def make_mask(self, ids, mask, n):
        self.lib.call("zk_make_mask", ids.data_ptr(), mask.data_ptr(), n, self.stream)

    # ---- average attention network (transformer_aan.py:92-117,186-189) ------------------
    def aan_fwd(self, x, mask, cat, B, L, H, use_mask):
        self.lib.call("zk_aan_fwd", x.ptr, mask.data_ptr(), cat.ptr, B, L, H, 1 if use_mask else 0, self.stream)

    def aan_bwd(self, dcat, dxg, dyg, ds, mask, dx, B, L, H, use_mask):
        self.lib.call("zk_aan_bwd", dcat.ptr, dxg.ptr, dyg.ptr, ds.ptr, mask.data_ptr(), dx.ptr, B, L, H,
                      1 if use_mask else 0, self.stream)

    # ---- merged attention of transformer_fuse (func.py:258-275) ------------------------------
    def cumavg_add_fwd(self, vq, mask, att, out, B, L, H):
        self.lib.call("zk_cumavg_add_fwd", vq.ptr, mask.data_ptr(), att.ptr, out.ptr, B, L, H, self.stream)

    def cumavg_bwd(self, dy, mask, dvq, B, L, H):
        self.lib.call("zk_cumavg_bwd", dy.ptr, mask.data_ptr(), dvq.ptr, B, L, H, self.stream)

    def aan_gate_fwd(self, z, cat, out, rows, H):
        self.lib.call("zk_aan_gate_fwd", z.ptr, cat.ptr, out.ptr, rows, H, self.stream)

    def aan_gate_bwd(self, dg, z, cat, dz, dxg, dyg, rows, H):
        self.lib.call("zk_aan_gate_bwd", dg.ptr, z.ptr, cat.ptr, dz.ptr, dxg.ptr, dyg.ptr, rows, H, self.stream)

    # ---- layer programs: a run of sentence-local ops as ONE persistent launch (zk_layer.hip) --------------
    def run_program(self, sentences, fn):
        """Issue the launches of ``fn()`` as one layer program when every one of them can be an op of it
        (zk_gemm with an untransposed A, zk_attn_fwd / zk_attn_bwd on the MFMA tiles, zk_add_ln_fwd), else as
        ordinary launches.  ``fn`` is run in recording mode first -- the entry points append ops instead of
        launching -- so the host-side schedule is written once (zero_amd/models/_core.py) for both forms.
        The device copy of the op table is cached by content: replays and hipGraph captures only launch."""
        if not self.programs_enabled or not self.lib.experiments:
            return fn()
        lib = self.lib
        op_bytes = lib.query("zk_prog_op_bytes")
        cap = 512
        if getattr(self, "_prog_host", None) is None:
            self._prog_host = ctypes.create_string_buffer(cap * op_bytes)
            self._prog_cache = {}
        lib.call("zk_prog_begin", int(sentences))
        lib.recording = True
        n, bwd = ctypes.c_int(0), ctypes.c_int(0)
        try:
            out = fn()
            ok = True
        except hip.ZeroHipError:
            ok = False
        finally:
            lib.recording = False
            rc = lib.raw("zk_prog_end")(self._prog_host, cap * op_bytes, ctypes.byref(n), ctypes.byref(bwd))
        if not ok or rc != 0 or n.value == 0:
            return fn()                      # nothing was launched while recording: issue it all normally
        key = (self._prog_host.raw[:n.value * op_bytes], int(sentences))
        ent = self._prog_cache.get(key)
        if ent is None:
            dev = torch.frombuffer(bytearray(key[0]), dtype=torch.uint8).to(self.device)
            state = torch.zeros(lib.query("zk_prog_state_bytes") // 4, dtype=torch.int32, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()      # the upload is complete before any capture
            ent = (dev, state)
            self._prog_cache[key] = ent
        lib.call("zk_prog_launch", ent[0].data_ptr(), n.value, int(sentences), bwd.value, ent[1].data_ptr(), self.stream)
        self.last_program_state = ent[1]
        return out

    def program_status(self):
        """(workgroups found off their XCD, aborted) of the last program launch -- forces a sync (tests)."""
        st = getattr(self, "last_program_state", None)
        if st is None:
            return None
        h = st.cpu()
        return int(h[576]), bool(h[577] != 0)

    # ---- hipGraph capture of a launch sequence ----------------------------------------
    @property
    def work_stream(self):
        """Persistent stream for captured steps (per-stream scratch is keyed by stream, so the eager
        sizing pass and the capture must run on the same one)."""
        if getattr(self, "_work_stream", None) is None:
            # High priority on a single rank: the upload stream's copies and id preparation of the NEXT batch run beside
            # the step and took ~45 us of it at equal priority (same-box A/B, rotating batches 4.265 -> 4.22 ms; a static
            # replay is unaffected).  With several ranks the gradient all-reduce runs on RCCL's own streams beside the
            # backward and must not be starved by it: normal priority there.  ZERO_HIP_WORK_PRIO overrides.
            prio = os.environ.get("ZERO_HIP_WORK_PRIO")
            if prio is None:
                multi = torch.distributed.is_available() and torch.distributed.is_initialized() and \
                    torch.distributed.get_world_size() > 1
                prio = 0 if multi else -1
            self._work_stream = torch.cuda.Stream(self.device, priority=int(prio))
        return self._work_stream

    def graph_capture(self, fn):
        """Capture ``fn()``'s launches on the CURRENT stream, return a replayable handle."""
        s = torch.cuda.current_stream(self.device)
        self.lib.call("zk_graph_begin", s.cuda_stream)
        try:
            fn()
        finally:
            exec_ = ctypes.c_void_p()
            self.lib.call("zk_graph_end", s.cuda_stream, ctypes.byref(exec_))
        self.last_graph_nodes = int(self.lib.raw("zk_graph_last_nodes")())
        return exec_

    def graph_launch(self, exec_):
        self.lib.call("zk_graph_launch", exec_, self.stream)
