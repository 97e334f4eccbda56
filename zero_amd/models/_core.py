# coding: utf-8
"""Hand-scheduled forward + backward of the Transformer family on HIP kernels.

One class drives the three registered models (``transformer``,
``transformer_aan``, ``transformer_rpr``): the same layer schedule as the
reference's ``encoder`` / ``decoder`` (models/transformer.py:15-218,
models/transformer_aan.py:92-260, models/transformer_rpr.py) but issued eagerly
as HIP launches on one stream -- there is no graph builder and no autodiff, the
backward below is written by hand against the same saved activations.

Saved for the backward, per sub-layer: the bf16 input of every GEMM, the
pre-LayerNorm sum with its fp32 mean / rstd, the attention log-sum-exp.
Everything lives in named persistent buffers (static addresses), so a whole
step can be captured into a hipGraph and replayed.
"""

import os

import numpy as np
import torch

from zero_amd import hip
from zero_amd.func import Engine, Mat
from zero_amd.variables import get_store

BF16 = torch.bfloat16
F32 = torch.float32


def trim_columns(ids):
    """utils/util.py:274-287 remove_invalid_seq on the host: drop columns where
    every row is pad (id 0), always keep column 0."""
    ids = np.asarray(ids)
    keep = (ids != 0).any(axis=0)
    if keep.size:
        keep[0] = True
    return np.ascontiguousarray(ids[:, keep])


class LazyLN(object):
    """LN(s) that no kernel has written (round 4): the un-normalised sum ``s`` a producer GEMM left (zk_gemm_ln), its
    per-64-column row statistics ``part`` and the LayerNorm scope whose gamma / beta apply.  Readers normalise where they
    read: the consumer GEMM algebraically, the next residual on the fly, the backward from the same statistics."""
    __slots__ = ("s", "part", "scope", "tag", "rows")

    def __init__(self, s, part, scope, tag):
        self.s, self.part, self.scope, self.tag, self.rows = s, part, scope, tag, s.rows


class TransformerCore(object):
    def __init__(self, params, model_name, store=None, device=None):
        self.hp = params
        self.model = model_name
        if device is None:
            device = "cuda:%d" % torch.cuda.current_device() if torch.cuda.is_available() else "cpu"
        self.eng = Engine(device)
        self.store = store if store is not None else get_store(params, model_name, device)
        self.H = params.hidden_size
        self.F = params.filter_size
        self.nh = params.num_heads
        self.d = self.H // self.nh
        self.rpr = model_name == "transformer_rpr"
        self.aan = model_name == "transformer_aan"
        # transformer_fuse (transformer_fuse.py:131-160): no decoder self-attention; one merged
        # sub-layer = cross-attention + averaged v_map(query) (func.py:258-275), then the FFN
        self.fuse = model_name == "transformer_fuse"
        self.cross = "fuse_attention" if self.fuse else "cross_attention"
        if self.aan and [s.lower() for s in params.strategies] != ["aan"]:
            raise NotImplementedError("Not supported: {}".format(params.strategies))
        shared = params.shared_source_target_embedding
        self.src_emb = "embedding" if shared else "src_embedding"
        self.tgt_emb = "embedding" if shared else "tgt_embedding"
        if shared:
            self.soft_emb = "embedding"
        else:
            self.soft_emb = "tgt_embedding" if params.shared_target_softmax_embedding else "softmax_embedding"
        self.V = params.tgt_vocab.size()
        self.Vpad = self.store.pshape[self.soft_emb][0]
        # weight-gradient GEMMs and bias column sums leave the critical path: they run on a second
        # HIP stream, concurrently with the dgrad chain (both are latency-bound at this size)
        # (measured in round 1: with one rank the second stream LOSES ~3 % -- cross-stream graph
        # edges cost more than the overlap buys once the small GEMMs are grouped -- so it is opt-in)
        self.use_side = os.environ.get("ZERO_HIP_SIDE_STREAM", "0") != "0" and self.eng.lib.experiments
        # weight-gradient GEMMs are deferred and launched as ONE grouped grid per `group_layers`
        # layers (each is far too small to fill 256 CUs on its own)
        self.group_wgrad = True
        # tile of the grouped weight-gradient launch: 128x256 (four waves with a 128x64 register tile each + four
        # producer waves, scripts/gemm_big_bench.py: 780 -> 856 TF on the decoder side incl. the logits problem)
        # one group per side of the model with a single rank (fewest launches); smaller groups with
        # data parallelism so that the gradient all-reduce of finished layers starts early
        import torch.distributed as _dist
        _multi = (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() > 1) or \
            os.environ.get("ZERO_HIP_FORCE_SEGMENTED", "0") != "0"      # (test hook: the multi-rank step with one rank)
        self.group_layers = int(os.environ.get("ZERO_HIP_GROUP_LAYERS", "2" if _multi else "6"))
        # Single rank (round 3): EVERY weight gradient of the step in ONE grouped launch of 256x256 tiles (7.8 KB staged
        # per MFLOP against 11.7 for 128x256).  The coarse tile needs the big group: 922 tiles on 256 CUs = 3.6 rounds,
        # while the encoder's 288 tiles alone would be 1.1 -- which is why the 256x256 tile lost inside the per-side
        # groups of round 2.  The bias gradients ride along as column sums by MFMA (gemm256_acc<.., CS>).
        self.group_all = not _multi
        # ("256x256n": the LDS-DMA pieces of a K step issued right behind the barrier instead of spread between the MFMA
        # groups -- same-box A/B of the whole step: 4.708 ms spread, 4.673 ms not, 4.80 ms for the round-2 grouping)
        # (round 4: the measured-and-lost alternatives -- per-side groups, the spread DMA issue, 128x128 -- lost their
        # switches; the 32-deep ring of the EXPERIMENTS build is selected with ZERO_HIP_WGRAD_TILE=256x256k32 there)
        self.wgrad_tile = (256, 256, 0) if self.group_all else (128, 256)
        if self.eng.lib.experiments and os.environ.get("ZERO_HIP_WGRAD_TILE", "").lower() == "256x256k32":
            self.wgrad_tile = (256, 256, "k32")
        self._pending_wgrads = []
        self._pending_colsums = []     # (dY Mat, bias-gradient view, private partial buffer)
        self._pending_lnred = []       # (partials, rows, H, dgamma, dbeta, dbias_prev)
        self._ln_bwd_done = {}         # tag -> (ds, dy) of a LayerNorm backward that ran inside a dgrad launch
        self._ln_next = None           # the LayerNorm below the sub-layer whose backward is being issued
        self._pending_rpr = []         # (partials address, slices, n, d rpr_k, d rpr_v): table gradients of the folded backward
        self._pending_adds = []        # (fp32 gradient view, fp32 temporary): dst += src after the flush
        self._mem_segs = []            # (dK or dV, W) pairs of the cross-attention memory side (see _finish_mem_grad)
        # encoder-output gradient by one K-segmented GEMM (needs the MFMA path: H a multiple of 64, aligned rows)
        self.kseg_mem = True
        # fused logits + cross entropy (no [T, V] fp32 logits in HBM, recompute in the backward): measured
        # 224 + 8 us forward and 287 us backward against 291 + 208 us for GEMM + k_ce_fused -- the second
        # pass over the 137-GFLOP GEMM costs what the saved 1 GB of traffic buys, so it is opt-in (it frees
        # T*V*4 bytes, which matters for larger batches / vocabularies)
        self.fused_ce = os.environ.get("ZERO_HIP_FUSED_CE", "0") != "0" and self.eng.lib.experiments
        self.logits_tile256 = True
        # the dgrad of the attention output projection inside the attention backward launch (zk_attn_bwd oproj_*): 18
        # launches and 18 [T, H] matrices less per step, same-box A/B -0.02 to -0.05 ms (the 64 x 64 x H product per
        # (sentence, head) costs the launch +7 us, the GEMM it replaces was 8-9 us).  By rule (_use_oproj): only while the
        # (sentence, head) workgroups fit the chip at once and H <= 512 -- with 2048 workgroups (256 sentences) or the big
        # widths (H = 1024: two 512-column chunks per workgroup, 1024 workgroups) the GEMM launch is the cheaper place for
        # the product (same-box: 12.39 vs 12.28 ms and 11.12 vs 11.01 ms).  Not with relative positions: the oproj variant
        # of that kernel runs one workgroup per CU and the longer prologue is not hidden (5.21 -> 5.24 ms).
        self.attn_oproj = os.environ.get("ZERO_HIP_ATTN_OPROJ", "auto").lower()
        self._red_id = 0
        self._side_stream = None      # created on first use: every stream of the process takes a share of the hardware queues
        # Residual + LayerNorm without a launch of its own in the TRAINING forward (round 4): the
        # o_map / ffn-output GEMM adds the residual and leaves the un-normalised sum + row statistics (zk_gemm_ln), the
        # next linear reads the sum against gamma-folded weights (zk_ln_fold, one launch per step), the next residual
        # normalises on the fly, and the LayerNorm backward -- which runs anyway -- writes the normalised rows for the
        # deferred weight gradients.  28 of the 30 LayerNorm-forward launches of a Transformer-base step go (the two that
        # end the encoder / decoder stacks keep their launch: their outputs feed grouped / 256-tile GEMMs).
        # MEASURED (profiles/r04_negative_results.txt): parity holds at the BASELINE sizes, the step does NOT get
        # faster (+0.03 to +0.09 ms) -- what the 28 launches did has to be done somewhere: the producer epilogues cost
        # +2 us each, the LayerNorm backward +5..7 us for the normalised rows it now writes, the fold launch 42 us.  So it
        # is an experiment: only in a `make EXPERIMENTS=1` library, only with ZERO_HIP_LAZY_LN=1.
        self.lazy_ln_mode = os.environ.get("ZERO_HIP_LAZY_LN", "0").lower() if self.eng.lib.experiments else "0"
        self.sync_ln_mode = os.environ.get("ZERO_HIP_SYNC_LN", "1") != "0"
        # round 6: small launches of the step merged pairwise (both input embeddings; their two gradient scatters; the two
        # bias column sums; per-sentence loss + mean): ZERO_HIP_MERGE_SMALL=0 restores the round-5 launches (A/B, tests)
        self.merge_small = os.environ.get("ZERO_HIP_MERGE_SMALL", "1") != "0"
        if self.eng.device.type == "cuda":
            self.eng.lib.raw("zk_tune")(16, 0 if self.merge_small else 1)      # (bit 0: per-sentence loss + mean as two launches)
        # (A/B: "noattn" keeps the attention forward a launch of its own, "fwd" also the LayerNorm backward)
        self.sync_ln_bwd = os.environ.get("ZERO_HIP_SYNC_LN", "1") != "fwd"
        self.sync_attn = os.environ.get("ZERO_HIP_SYNC_LN", "1") not in ("noattn", "fwd", "0")
        # round 6: the projection in front of the attention (merged qkv_map / q_map) as the prologue of that launch
        # (zk_proj_attn_out_ln); ZERO_HIP_PROJ_ATTN=0 keeps it a launch of its own (A/B, bit-identity tests)
        self.proj_attn = os.environ.get("ZERO_HIP_PROJ_ATTN", "1") != "0"
        # the attention BACKWARD inside the dgrad launch: measured, no gain (EXPERIMENTS=1 library, ZERO_HIP_SYNC_LN=attnbwd)
        self.sync_attn_bwd = os.environ.get("ZERO_HIP_SYNC_LN", "1") == "attnbwd" and self.eng.lib.experiments
        self._sync_ln = False
        # The update of the weight matrices inside the weight-gradient launch (round 4; zk_gemm_grouped_update): set by the
        # Trainer for a step whose update is norm-free, single-rank and unaccumulated; the backward's one grouped launch
        # then runs TF1 Adam on its accumulators for every weight it can (see _flush_wgrads) and `fused_info` tells the
        # optimiser what is left to do.
        self.fused_update = None      # dict(master, m, v, shadow, grad, hyper) or None
        self.fused_info = None        # (ranges, sq, n_extra) of the last backward, or None
        self._lazy = False            # set per forward()
        self._lazy_tags = {}          # tag -> LazyLN of this step's forward (read by the backward)

    def _use_lazy_ln(self, train, save):
        if self.lazy_ln_mode != "1" or not (train and save):
            return False
        ok = self.group_all and self.group_wgrad and self.eng.gemm_impl == 0 and not self.use_side and \
            not self.aan and not self.fuse and self.H % 128 == 0 and self.H // 64 <= 16 and self.F % 64 == 0 and \
            not self.eng.programs_enabled
        if self.lazy_ln_mode == "1" and not ok:
            raise hip.ZeroHipError("ZERO_HIP_LAZY_LN=1: the LayerNorm-free forward needs one rank (all weight gradients in "
                                   "one deferred launch), the plain / rpr Transformer and H a multiple of 128 (<= 1024)")
        return ok

    def _lazy_pairs(self):
        """[(linear scope, LayerNorm scope)]: every linear layer of the training forward whose input is the output of a
        LayerNorm that is not materialised -- the weights zk_ln_fold prepares at the head of the step."""
        hp, out = self.hp, []
        for l in range(hp.num_encoder_layer):
            pre = "encoder/layer_%d" % l
            if l > 0:
                out.append((pre + "/self_attention/dot_attention/qkv_map", "encoder/layer_%d/feed_forward" % (l - 1)))
            out.append((pre + "/feed_forward/ffn_layer/enlarge", pre + "/self_attention"))
        for l in range(hp.num_decoder_layer):
            pre = "decoder/layer_%d" % l
            if l > 0:
                out.append((pre + "/self_attention/dot_attention/qkv_map", "decoder/layer_%d/feed_forward" % (l - 1)))
            out.append((pre + "/%s/dot_attention/q_map" % self.cross, pre + "/self_attention"))
            out.append((pre + "/feed_forward/ffn_layer/enlarge", pre + "/" + self.cross))
        return out

    def _fold_weights(self):
        """zk_ln_fold for every pair of _lazy_pairs(): ONE launch at the head of the step (the fp32 masters changed in
        the previous update)."""
        e, st = self.eng, self.store
        probs = []
        for lin, ln in self._lazy_pairs():
            W = st.w(lin + "/W_0_0")
            K, N = W.shape
            probs.append((W, st.w(ln + "/layer_norm/scale"), st.w(ln + "/layer_norm/offset"), st.w(lin + "/b_0"),
                          e.mat("fold.W." + lin, K, N), e.buf("fold.c." + lin, (N,), F32), e.buf("fold.d." + lin, (N,), F32)))
        e.ln_fold(probs)

    def _use_oproj(self, B):
        if self.attn_oproj in ("0", "1"):
            return self.attn_oproj == "1"
        return (not self.rpr) and self.H <= 512 and B * self.nh <= 512

    @property
    def side(self):
        if self._side_stream is None and self.eng.device.type == "cuda":
            self._side_stream = torch.cuda.Stream(self.eng.device)
        return self._side_stream

    # ------------------------------------------------------------------ stream plumbing
    def _side(self, fn):
        """Run fn() on the side stream, ordered after everything enqueued so far on the current one."""
        if not self.use_side:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()

    def _flush_wgrads(self):
        """Launch every deferred weight-gradient GEMM as one grouped grid on the side stream."""
        if self._pending_wgrads:
            probs = self._pending_wgrads
            self._pending_wgrads = []
            if self.fused_update is not None and self.group_all and self.wgrad_tile == (256, 256, 0) and \
                    not self.use_side and self.eng.gemm_impl == 0 and self.fused_info is None and \
                    not self._pending_adds:      # (a variable used twice gets a second contribution AFTER this launch)
                self.fused_info = self.eng.gemm_grouped_update(probs, self.fused_update)
            else:
                self._side(lambda: self.eng.gemm_grouped([p[:9] for p in probs], 1, 0, tile=self.wgrad_tile))
        if self._pending_colsums or self._pending_lnred or self._pending_rpr:
            cs, ln, rp = self._pending_colsums, self._pending_lnred, self._pending_rpr
            self._pending_colsums, self._pending_lnred, self._pending_rpr = [], [], []
            self._side(lambda: self.eng.reductions_grouped(cs, ln, rp))
        if self._pending_adds:
            adds = self._pending_adds
            self._pending_adds = []
            self._side(lambda: [self._accumulate(dst, src) for dst, src in adds])

    def _defer_rpr(self):
        """Relative positions: the table-gradient partials of every attention layer summed by the grouped reduction
        launch of the layer group (18 launches fewer per step than one reduction per attention)."""
        return self.rpr and self.group_wgrad and self.eng.gemm_impl == 0

    def _use_kseg(self):
        return self.kseg_mem and self.H % 64 == 0 and self.eng.gemm_impl == 0

    def _finish_mem_grad(self, d_mem):
        """d_mem = sum over decoder layers of dK_l W_k,l^T + dV_l W_v,l^T: one zk_gemm_kseg launch per 16 segments
        (12 segments of K = H for six layers) instead of a chain of 2 L in-place GEMMs."""
        segs, self._mem_segs = self._mem_segs, []
        first = True
        for i in range(0, len(segs), 16):
            self.eng.gemm_kseg(segs[i:i + 16], d_mem, d_mem.rows, self.H, self.H, 1,
                               residual=None if first else d_mem)
            first = False

    def _accumulate(self, dst, src):
        """dst += src (fp32 gradient views of equal size)."""
        self.eng.lib.call("zk_axpby_f32", dst.data_ptr(), src.data_ptr(), 1.0, 1.0, dst.numel(), self.eng.stream)

    def _join_side(self):
        if self.use_side:
            ev = torch.cuda.Event()
            ev.record(self.side)
            torch.cuda.current_stream(self.eng.device).wait_event(ev)

    # ------------------------------------------------------------------ helpers
    def W(self, name):
        t = self.store.s(name)
        return Mat(t, t.shape[0], t.shape[1])

    def b(self, name):
        return self.store.w(name)

    def gW(self, name):
        t = self.store.g(name)
        return Mat(t, t.shape[0], t.shape[1])

    def gb(self, name):
        return self.store.g(name)

    def _linear(self, x, scope, out, act=0, drop_p=0.0, sid=0):
        """y = x @ W + b  (func.py:14-65).  x a LazyLN: y = LN(s) @ W + b from the un-normalised sum, the gamma-folded
        weight and the row statistics (zk_gemm_ln consumer form)."""
        if isinstance(x, LazyLN):
            e = self.eng
            K, N = self.store.pshape[scope + "/W_0_0"]
            e.gemm_ln(x.s, e.mat("fold.W." + scope, K, N), out, x.rows, N, K, e.buf("fold.d." + scope, (N,), F32),
                      self.H // 64, act=act, drop_p=drop_p, sid=sid, in_part=x.part,
                      in_c=e.buf("fold.c." + scope, (N,), F32))
            return
        Wm = self.W(scope + "/W_0_0")
        self.eng.gemm(x, Wm, out, x.rows, Wm.cols, Wm.rows, 0, 0, bias=self.b(scope + "/b_0"), act=act,
                      drop_p=drop_p, sid=sid)

    def _linear_bwd(self, x, dy, scope, dx=None, residual=None, bias_grad=True, act=0, aux=None, aux_scale=1.0,
                    accumulate=False, ln_next=None):
        """dW = x^T dy (fp32, overwrite), db = colsum(dy), dx = dy @ W^T (+residual).
        accumulate: the variable is used twice in the graph (v_map of the merged attention); this
        use's gradients go to temporaries that are added once the overwriting use has run."""
        Wm = self.W(scope + "/W_0_0")
        gW, gb = self.gW(scope + "/W_0_0"), self.gb(scope + "/b_0")
        adds = []
        if accumulate:
            tW = self.eng.buf("g.acc.W." + scope, tuple(gW.t.shape), F32)
            tb = self.eng.buf("g.acc.b." + scope, tuple(gb.shape), F32)
            adds = [(gW.t, tW)] + ([(gb, tb)] if bias_grad else [])
            gW, gb = Mat(tW, gW.rows, gW.cols), tb

        # bias gradient = column sums of dY: on the wide (producer-wave) tiles the producers of the weight-gradient GEMM
        # compute them from the dY tiles they stage anyway (no separate pass over dY)
        # (256x256 tiles: by two extra MFMAs per eight on the fragments the tm = 0 tiles hold anyway)
        fold_cs = bias_grad and isinstance(self.wgrad_tile, tuple)
        if self.group_wgrad and self.eng.gemm_impl == 0:
            # (10th element: the variable's gradient is complete with this product -> its update may run in the launch)
            self._pending_wgrads.append((x, dy, gW, Wm.rows, Wm.cols, x.rows, None, None, gb if fold_cs else None,
                                         not accumulate and not adds))
            if bias_grad and not fold_cs:
                gy = self.eng.lib.raw("zk_colsum_rowchunks")(dy.rows)
                pw = self.eng.buf("g.cs%d" % len(self._pending_colsums) + scope, (gy * dy.cols,), F32)
                self._pending_colsums.append((dy, gb, pw))
            self._pending_adds += adds
        else:
            def wgrad():
                self.eng.gemm(x, dy, gW, Wm.rows, Wm.cols, x.rows, 1, 0)
                if bias_grad:
                    self.eng.colsum(dy, gb)
            self._side(wgrad)
            # the overwriting use of the variable must already have run: callers issue it first
            self._pending_adds += adds
        if dx is not None:
            if ln_next is not None and act == 0 and residual is not None and self._sync_fits(dy.rows):
                # dx is only read by the backward of the LayerNorm below this sub-layer: that backward runs in the
                # epilogue of this product and dx is never written (zk_gemm_ln_bwd)
                self._gemm_ln_bwd(dy, Wm, residual, ln_next)
                return
            self.eng.gemm(dy, Wm, dx, dy.rows, Wm.rows, Wm.cols, 0, 1, residual=residual, act=act, aux=aux,
                          aux_scale=aux_scale)

    def _attn_bwd_ln(self, q, k, v, att, lse, dq, dk, dv, B, Lq, Lk, kmask, causal, attn_sid, oproj, dA, Wm, residual):
        """attention backward + the dgrad of the projection in front of it + the LayerNorm backward below in one launch
        (zk_attn_bwd_ln) when self._ln_next names that LayerNorm and the shape is covered; False otherwise (the caller
        issues attn_bwd and the dgrad)."""
        e, H, hp = self.eng, self.H, self.hp
        if not (self._ln_next is not None and self.sync_attn_bwd and oproj is not None and not self.rpr and self.d == 64 and
                Lq <= 64 and Lk <= 64 and e.attn_impl in (0, 2)):
            return False
        scope, tag, prev_bias, drop_p, sid = self._ln_next
        T = B * Lq
        ds = e.mat("g.%s.ds" % tag, T, H)
        dyo = e.mat("g.%s.dy" % tag, T, H) if drop_p > 0.0 else None
        nbytes = max(B * 3 * H * 4, e.lib.query("zk_gemm_ln_bwd_partials", T, H), e.lib.query("zk_add_ln_bwd_workspace", T, H))
        pws = e.buf("g.%s.lnws" % tag, (nbytes // 4,), F32)
        ok = e.attn_bwd_ln(q, k, v, att, lse, dq, dk, dv, B, self.nh, Lq, Lk, self.d, kmask, causal, hp.attention_dropout,
                           attn_sid, oproj, dA, Wm, residual, e.mat(tag + ".s", T, H), e.buf(tag + ".mean", (T,), F32),
                           e.buf(tag + ".rstd", (T,), F32), self.b(scope + "/layer_norm/scale"), ds, dyo, pws, drop_p, sid)
        if not ok:
            return False
        dbp = self.gb(prev_bias) if prev_bias is not None else None
        self._pending_lnred.append((pws, T, H, self.gb(scope + "/layer_norm/scale"), self.gb(scope + "/layer_norm/offset"),
                                    dbp, B))
        self._ln_bwd_done[tag] = (ds, dyo if dyo is not None else ds)
        return True

    def _gemm_ln_bwd(self, dy, Wm, residual, ln_next):
        e, H = self.eng, self.H
        scope, tag, prev_bias, drop_p, sid = ln_next
        T = dy.rows
        ds = e.mat("g.%s.ds" % tag, T, H)
        dyo = e.mat("g.%s.dy" % tag, T, H) if drop_p > 0.0 else None
        nbytes = max(e.lib.query("zk_gemm_ln_bwd_partials", T, H), e.lib.query("zk_add_ln_bwd_workspace", T, H))
        pws = e.buf("g.%s.lnws" % tag, (nbytes // 4,), F32)
        e.gemm_ln_bwd(dy, Wm, T, H, Wm.cols, residual, e.mat(tag + ".s", T, H), e.buf(tag + ".mean", (T,), F32),
                      e.buf(tag + ".rstd", (T,), F32), self.b(scope + "/layer_norm/scale"), ds, dyo, pws, drop_p, sid)
        dbp = self.gb(prev_bias) if prev_bias is not None else None
        self._pending_lnred.append((pws, T, H, self.gb(scope + "/layer_norm/scale"), self.gb(scope + "/layer_norm/offset"),
                                    dbp, True))
        self._ln_bwd_done[tag] = (ds, dyo if dyo is not None else ds)

    # ------------------------------------------------------------------ sub-layers (forward)
    def _out_ln(self, a, lin, x, scope, tag, save, drop_p, sid, last=False):
        """The tail of a sub-layer: LN(x + dropout(a @ W + b)) (func.py:321-324, 289-303; transformer.py:57-58).
        Default: the GEMM, then the residual + LayerNorm launch.  With the LayerNorm-free forward (self._lazy) the GEMM
        adds the residual itself -- normalising it on the fly when x is a LazyLN -- and leaves the un-normalised sum with
        its row statistics; nothing normalises it here unless it ends a stack (last)."""
        e, H = self.eng, self.H
        T = a.rows
        if not self._lazy:
            if self._sync_ln and not e.lib.recording and self._sync_fits(T):
                # one launch: the workgroups of a block of rows exchange their LayerNorm partials (zk_gemm_add_ln)
                Wm = self.W(lin + "/W_0_0")
                out = e.mat(tag + ".o", T, H)
                s = e.mat(tag + ".s", T, H) if save else None
                mean = e.buf(tag + ".mean", (T,), F32) if save else None
                rstd = e.buf(tag + ".rstd", (T,), F32) if save else None
                e.gemm_add_ln(a, Wm, T, H, Wm.rows, self.b(lin + "/b_0"), x, self.b(scope + "/layer_norm/scale"),
                              self.b(scope + "/layer_norm/offset"), out, s, mean, rstd, drop_p, sid)
                return out
            y = e.mat("tmp.y%d" % T, T, H)
            self._linear(a, lin, y)
            return self._ln_fwd(x, y, scope, tag, save, drop_p, sid)
        Wm = self.W(lin + "/W_0_0")
        s = e.mat(tag + ".s", T, H)
        part = e.buf(tag + ".part", (T, H // 64, 2), F32)
        lazy_res = isinstance(x, LazyLN)
        e.gemm_ln(a, Wm, s, T, H, Wm.rows, self.b(lin + "/b_0"), H // 64, residual=x.s if lazy_res else x,
                  drop_p=drop_p, sid=sid, stat_out=part, res_part=x.part if lazy_res else None,
                  res_gamma=self.b(x.scope + "/layer_norm/scale") if lazy_res else None,
                  res_beta=self.b(x.scope + "/layer_norm/offset") if lazy_res else None)
        if not last:
            out = LazyLN(s, part, scope, tag)
            self._lazy_tags[tag] = out
            return out
        # end of a stack: the normalised rows feed grouped / 256-tile GEMMs -> one LayerNorm launch over the stored sum
        # (its own statistics; the backward of this sub-layer is the ordinary one)
        out = e.mat(tag + ".o", T, H)
        e.add_ln_fwd(s, None, self.b(scope + "/layer_norm/scale"), self.b(scope + "/layer_norm/offset"), out, None,
                     e.buf(tag + ".mean", (T,), F32), e.buf(tag + ".rstd", (T,), F32), 0.0, sid)
        return out

    def _sync_fits(self, rows):
        """The in-launch exchange pays while the launch is little more than ONE resident round of workgroups (two 64x64-tile
        workgroups per CU): a workgroup that waits for its peers keeps its slot, and with several rounds the waiting slots
        are taken from workgroups that could run.  Measured, ms per step with / without it: 2560 rows x 512: 3.80 / 4.03;
        4096 x 512 (512 workgroups): 4.26 / 4.55; 6144 x 512 (768): 6.17 / 6.34; 8192 x 512 (1024): 7.26 / 7.25; 16384 x 512:
        13.15 / 12.22; 4096 x 1024 (1024, sixteen peers): 10.63 / 10.51.  Rule: at most three workgroups per CU."""
        return ((rows + 63) // 64) * (self.H // 64) <= 3 * self.eng.cu_count

    def _attn_out_ln_ok(self, B, Lq, Lk):
        e = self.eng
        return (self._sync_ln and self.sync_attn and not self.rpr and not e.lib.recording and self.d == 64 and
                Lq <= 64 and Lk <= 256 and e.attn_impl in (0, 2) and B * self.nh <= 3 * e.cu_count)

    def _attn_out_ln(self, q, k, v, att, lse, B, Lq, Lk, kmask, causal, attn_drop, attn_sid, lin, x, scope, tag, save,
                     drop_p, sid, proj=None):
        """attention + o_map + residual + LayerNorm in one launch when the in-launch LayerNorm is on and the shape is
        covered (zk_attn_out_ln: no relative positions, Lq <= 64, Lk <= 256, 64-wide heads); None otherwise.
        proj = (scope of the linear in front, 1 | 3): q (k, v) = x W + b is computed inside the launch as well."""
        e, H = self.eng, self.H
        if not self._attn_out_ln_ok(B, Lq, Lk):
            return None
        T = x.rows
        Wm = self.W(lin + "/W_0_0")
        out = e.mat(tag + ".o", T, H)
        s = e.mat(tag + ".s", T, H) if save else None
        mean = e.buf(tag + ".mean", (T,), F32) if save else None
        rstd = e.buf(tag + ".rstd", (T,), F32) if save else None
        pj = None
        if proj is not None:
            pj = (x, self.W(proj[0] + "/W_0_0"), self.b(proj[0] + "/b_0"), proj[1])
        ok = e.attn_out_ln(q, k, v, att, lse, B, self.nh, Lq, Lk, self.d, kmask, causal, attn_drop, attn_sid, Wm,
                           self.b(lin + "/b_0"), x, self.b(scope + "/layer_norm/scale"), self.b(scope + "/layer_norm/offset"),
                           out, s, mean, rstd, drop_p, sid, proj=pj)
        return out if ok else None

    def _ln_fwd(self, x, y, scope, tag, save, drop_p, sid):
        T, H = x.rows, self.H
        e = self.eng
        out = e.mat(tag + ".o", T, H)
        if save:
            s = e.mat(tag + ".s", T, H)
            mean = e.buf(tag + ".mean", (T,), F32)
            rstd = e.buf(tag + ".rstd", (T,), F32)
        else:
            s, mean, rstd = None, None, None
        e.add_ln_fwd(x, y, self.b(scope + "/layer_norm/scale"), self.b(scope + "/layer_norm/offset"), out,
                     s, mean, rstd, drop_p, sid)
        return out

    def _self_attn_fwd(self, x, B, L, scope, tag, kmask, causal, save, sid0, train):
        e, H = self.eng, self.H
        hp = self.hp
        T = x.rows
        p = scope + "/dot_attention/"
        qkv = e.mat(tag + ".qkv", T, 3 * H)
        att = e.mat(tag + ".att", T, H)
        lse = e.buf(tag + ".lse", (B * self.nh * L,), F32) if save else None
        rk = self.store.s(p + "rpr_keys/embeddings") if self.rpr else None
        rv = self.store.s(p + "rpr_values/embeddings") if self.rpr else None
        fuse_proj = self.proj_attn and not isinstance(x, LazyLN) and self._attn_out_ln_ok(B, L, L)
        for pj in ((p + "qkv_map", 3), None) if fuse_proj else (None,):
            if pj is None:
                self._linear(x, p + "qkv_map", qkv)
            out = self._attn_out_ln(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att, lse, B,
                                    L, L, kmask, causal, hp.attention_dropout if train else 0.0, sid0, p + "o_map", x, scope,
                                    tag, save, hp.residual_dropout if train else 0.0, sid0 + 1, proj=pj)
            if out is not None:
                return out
        e.attn_fwd(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att, lse, B,
                   self.nh, L, L, self.d, kmask=kmask, causal=causal, rpr_k=rk, rpr_v=rv,
                   max_rel=hp.max_relative_position, drop_p=hp.attention_dropout if train else 0.0, sid=sid0)
        return self._out_ln(att, p + "o_map", x, scope, tag, save, hp.residual_dropout if train else 0.0, sid0 + 1)

    def _cross_kv_grouped(self, mem, n_layers):
        """k_map / v_map of EVERY decoder layer read only the encoder output (func.py:206-216): one
        grouped GEMM on the side stream, overlapped with the first decoder sub-layer."""
        e, H = self.eng, self.H
        probs = []
        for l in range(n_layers):
            p = "decoder/layer_%d/%s/dot_attention/" % (l, self.cross)
            kv = e.mat("d%d.ca.kv" % l, mem.rows, 2 * H)
            for nm, c0 in (("k_map", 0), ("v_map", H)):
                Wm = self.W(p + nm + "/W_0_0")
                probs.append((mem, Wm, kv.cols_slice(c0, c0 + H), mem.rows, H, H, self.b(p + nm + "/b_0")))
        self._side(lambda: e.gemm_grouped(probs, 0, 0))

    def _cross_attn_fwd(self, x, mem, B, Lq, Lk, scope, tag, kmask, save, sid0, train, kv_ready=False,
                        fuse_tmask=None):
        e, H = self.eng, self.H
        hp = self.hp
        p = scope + "/dot_attention/"
        q = e.mat(tag + ".q", x.rows, H)
        fuse_proj = (self.proj_attn and fuse_tmask is None and not isinstance(x, LazyLN) and
                     self._attn_out_ln_ok(B, Lq, Lk))
        if not fuse_proj:
            self._linear(x, p + "q_map", q)
        kv = e.mat(tag + ".kv", mem.rows, 2 * H)
        if not kv_ready:
            self._linear(mem, p + "k_map", kv.cols_slice(0, H))
            self._linear(mem, p + "v_map", kv.cols_slice(H, 2 * H))
        att = e.mat(tag + ".att", x.rows, H)
        lse = e.buf(tag + ".lse", (B * self.nh * Lq,), F32) if save else None
        rk = self.store.s(p + "rpr_keys/embeddings") if self.rpr else None
        rv = self.store.s(p + "rpr_values/embeddings") if self.rpr else None
        if fuse_tmask is None:
            for pj in ((p + "q_map", 1), None) if fuse_proj else (None,):
                if pj is None and fuse_proj:
                    self._linear(x, p + "q_map", q)           # the fused launch refused the shape
                out = self._attn_out_ln(q, kv.cols_slice(0, H), kv.cols_slice(H, 2 * H), att, lse, B, Lq, Lk, kmask, False,
                                        hp.attention_dropout if train else 0.0, sid0, p + "o_map", x, scope, tag, save,
                                        hp.residual_dropout if train else 0.0, sid0 + 1, proj=pj)
                if out is not None:
                    return out
        e.attn_fwd(q, kv.cols_slice(0, H), kv.cols_slice(H, 2 * H), att, lse, B, self.nh, Lq, Lk, self.d,
                   kmask=kmask, causal=False, rpr_k=rk, rpr_v=rv, max_rel=hp.max_relative_position,
                   drop_p=hp.attention_dropout if train else 0.0, sid=sid0)
        if fuse_tmask is not None:
            # func.py:258-275: o += average over valid positions j <= i of v_map(query)
            vq = e.mat("tmp.vq%d" % x.rows, x.rows, H)
            self._linear(x, p + "v_map", vq)
            atts = e.mat(tag + ".atts", x.rows, H)
            e.cumavg_add_fwd(vq, fuse_tmask, att, atts, B, Lq, H)
            att = atts
        return self._out_ln(att, p + "o_map", x, scope, tag, save, hp.residual_dropout if train else 0.0, sid0 + 1)

    def _ffn_fwd(self, x, scope, tag, save, sid0, train, last=False):
        e, H, F = self.eng, self.H, self.F
        hp = self.hp
        p = scope + "/ffn_layer/"
        h = e.mat(tag + ".h", x.rows, F)
        self._linear(x, p + "enlarge", h, act=1, drop_p=hp.relu_dropout if train else 0.0, sid=sid0)
        return self._out_ln(h, p + "output", x, scope, tag, save, hp.residual_dropout if train else 0.0, sid0 + 1,
                            last=last)

    def _aan_fwd(self, x, B, L, scope, tag, tmask, save, sid0, train):
        e, H = self.eng, self.H
        hp = self.hp
        T = x.rows
        cat = e.mat(tag + ".cat", T, 2 * H)
        e.aan_fwd(x, tmask, cat, B, L, H, hp.aan_mask)
        if hp.use_ffn:
            # transformer_aan.py:176-183: y = ffn_layer(average) (no residual / LN of its own)
            ya = e.mat(tag + ".ya", T, H)
            e.lib.call("zk_gather_rows", cat.ptr + H * 2, 2 * H * 2, None, ya.ptr, H * 2, T, H * 2, e.stream)
            h = e.mat(tag + ".h", T, self.F)
            self._linear(ya, scope + "/ffn_layer/enlarge", h, act=1, drop_p=hp.relu_dropout if train else 0.0,
                         sid=sid0 + 2)
            self._linear(h, scope + "/ffn_layer/output", cat.cols_slice(H, 2 * H))
        z = e.mat(tag + ".z", T, 2 * H)
        self._linear(cat, scope + "/z_project", z)
        g = e.mat("tmp.y%d" % T, T, H)
        e.aan_gate_fwd(z, cat, g, T, H)
        return self._ln_fwd(x, g, scope, tag, save, hp.residual_dropout if train else 0.0, sid0 + 1)

    # ------------------------------------------------------------------ sub-layers (backward)
    def _ln_bwd(self, dx, scope, tag, prev_bias, drop_p, sid, side):
        """returns (ds, dy): grads of the residual input and of the sub-layer output."""
        e, H = self.eng, self.H
        T = dx.rows
        done = self._ln_bwd_done.pop(tag, None)
        if done is not None:          # ran inside the dgrad launch that produced dx (_linear_bwd(ln_next=...))
            return done
        ds = e.mat("g.%s.ds" % tag, T, H)
        dy = e.mat("g.%s.dy" % tag, T, H) if drop_p > 0.0 else None
        dgam, dbet = self.gb(scope + "/layer_norm/scale"), self.gb(scope + "/layer_norm/offset")
        dbp = self.gb(prev_bias) if prev_bias is not None else None
        lz = self._lazy_tags.get(tag)
        if lz is not None:
            # the LayerNorm the forward never launched: statistics from the producer GEMM's partials; the normalised rows
            # (X operand of the consumer's deferred weight gradient) are written here, by the launch that runs anyway
            nbytes = e.lib.query("zk_add_ln_bwd_workspace", T, H)
            pws = e.buf("g.%s.lnws" % tag, (nbytes // 4,), F32)
            e.add_ln_bwd_lazy(dx, lz.s, lz.part, self.b(scope + "/layer_norm/scale"), self.b(scope + "/layer_norm/offset"),
                              e.mat(tag + ".o", T, H), ds, dy, dgam, dbet, dbp, drop_p, sid, private_ws=pws)
            self._pending_lnred.append((pws, T, H, dgam, dbet, dbp))
        elif self.group_wgrad and e.gemm_impl == 0:
            # per-block partial sums go to a buffer private to this sub-layer; the column reduction
            # joins the grouped reduction launch of this layer group
            nbytes = e.lib.query("zk_add_ln_bwd_workspace", T, H)
            pws = e.buf("g.%s.lnws" % tag, (nbytes // 4,), F32)
            e.add_ln_bwd(dx, e.mat(tag + ".s", T, H), e.buf(tag + ".mean", (T,), F32),
                         e.buf(tag + ".rstd", (T,), F32), self.b(scope + "/layer_norm/scale"), ds, dy, dgam, dbet,
                         dbp, drop_p, sid, private_ws=pws)
            self._pending_lnred.append((pws, T, H, dgam, dbet, dbp))
        else:
            e.add_ln_bwd(dx, e.mat(tag + ".s", T, H), e.buf(tag + ".mean", (T,), F32),
                         e.buf(tag + ".rstd", (T,), F32), self.b(scope + "/layer_norm/scale"), ds, dy, dgam, dbet,
                         dbp, drop_p, sid)
        return ds, (dy if dy is not None else ds)

    def _ffn_bwd(self, dx, x_in, scope, tag, sid0, side, dx_out):
        e, hp = self.eng, self.hp
        p = scope + "/ffn_layer/"
        T = dx.rows
        ds, dy = self._ln_bwd(dx, scope, tag, p + "output/b_0", hp.residual_dropout, sid0 + 1, side)
        h = e.mat(tag + ".h", T, self.F)
        dh = e.mat("g.%s.dh" % tag, T, self.F)
        rp = hp.relu_dropout
        self._linear_bwd(h, dy, p + "output", dx=dh, bias_grad=False, act=2, aux=h,
                         aux_scale=1.0 / (1.0 - rp) if rp > 0 else 1.0)
        self._linear_bwd(x_in, dh, p + "enlarge", dx=dx_out, residual=ds, ln_next=self._ln_next)
        return dx_out

    def _self_attn_bwd(self, dx, x_in, B, L, scope, tag, kmask, causal, sid0, side, dx_out):
        e, hp, H = self.eng, self.hp, self.H
        p = scope + "/dot_attention/"
        T = dx.rows
        ds, dy = self._ln_bwd(dx, scope, tag, p + "o_map/b_0", hp.residual_dropout, sid0 + 1, side)
        att = e.mat(tag + ".att", T, H)
        datt = e.mat("g.%s.datt" % tag, T, H)
        # the dgrad of o_map runs INSIDE the attention backward (its 64 x 64 x H piece per sentence and head): only the
        # weight gradient is recorded here
        oproj = (dy, self.W(p + "o_map/W_0_0")) if self._use_oproj(B) else None
        self._linear_bwd(att, dy, p + "o_map", dx=None if oproj else datt, bias_grad=False)
        qkv = e.mat(tag + ".qkv", T, 3 * H)
        dqkv = e.mat("g.%s.dqkv" % tag, T, 3 * H)
        rk = self.store.s(p + "rpr_keys/embeddings") if self.rpr else None
        rv = self.store.s(p + "rpr_values/embeddings") if self.rpr else None
        if self._attn_bwd_ln(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att,
                             e.buf(tag + ".lse", (B * self.nh * L,), F32), dqkv.cols_slice(0, H), dqkv.cols_slice(H, 2 * H),
                             dqkv.cols_slice(2 * H, 3 * H), B, L, L, kmask, causal, sid0, oproj, dqkv,
                             self.W(p + "qkv_map/W_0_0"), ds):
            self._linear_bwd(x_in, dqkv, p + "qkv_map", dx=None)          # the weight gradient only
            return dx_out
        e.attn_bwd(qkv.cols_slice(0, H), qkv.cols_slice(H, 2 * H), qkv.cols_slice(2 * H, 3 * H), att, datt,
                   e.buf(tag + ".lse", (B * self.nh * L,), F32), dqkv.cols_slice(0, H), dqkv.cols_slice(H, 2 * H),
                   dqkv.cols_slice(2 * H, 3 * H), B, self.nh, L, L, self.d, kmask=kmask, causal=causal,
                   rpr_k=rk, rpr_v=rv,
                   drpr_k=self.gb(p + "rpr_keys/embeddings") if self.rpr else None,
                   drpr_v=self.gb(p + "rpr_values/embeddings") if self.rpr else None,
                   max_rel=hp.max_relative_position, drop_p=hp.attention_dropout, sid=sid0,
                   defer_tables=(self._pending_rpr, tag) if self._defer_rpr() else None, oproj=oproj)
        self._linear_bwd(x_in, dqkv, p + "qkv_map", dx=dx_out, residual=ds, ln_next=self._ln_next)
        return dx_out

    def _cross_attn_bwd(self, dx, x_in, mem, d_mem, B, Lq, Lk, scope, tag, kmask, sid0, side, dx_out,
                        fuse_tmask=None):
        e, hp, H = self.eng, self.hp, self.H
        p = scope + "/dot_attention/"
        T = dx.rows
        ds, dy = self._ln_bwd(dx, scope, tag, p + "o_map/b_0", hp.residual_dropout, sid0 + 1, side)
        att = e.mat(tag + ".att", T, H)
        datt = e.mat("g.%s.datt" % tag, T, H)
        # merged attention: o_map saw att + averaged v_map(query); both terms get the same gradient (which the averaging's
        # backward reads too: there the product is formed by the GEMM as before)
        oproj = (dy, self.W(p + "o_map/W_0_0")) if self._use_oproj(B) and fuse_tmask is None else None
        self._linear_bwd(e.mat(tag + ".atts", T, H) if fuse_tmask is not None else att, dy, p + "o_map",
                         dx=None if oproj else datt, bias_grad=False)
        q = e.mat(tag + ".q", T, H)
        kv = e.mat(tag + ".kv", mem.rows, 2 * H)
        dq = e.mat("g.%s.dq" % tag, T, H)
        dkv = e.mat("g.%s.dkv" % tag, mem.rows, 2 * H)
        rk = self.store.s(p + "rpr_keys/embeddings") if self.rpr else None
        rv = self.store.s(p + "rpr_values/embeddings") if self.rpr else None
        if fuse_tmask is None and self._attn_bwd_ln(q, kv.cols_slice(0, H), kv.cols_slice(H, 2 * H), att,
                                                    e.buf(tag + ".lse", (B * self.nh * Lq,), F32), dq, dkv.cols_slice(0, H),
                                                    dkv.cols_slice(H, 2 * H), B, Lq, Lk, kmask, False, sid0, oproj, dq,
                                                    self.W(p + "q_map/W_0_0"), ds):
            self._linear_bwd(x_in, dq, p + "q_map", dx=None)              # the weight gradient only
        else:
            e.attn_bwd(q, kv.cols_slice(0, H), kv.cols_slice(H, 2 * H), att, datt,
                       e.buf(tag + ".lse", (B * self.nh * Lq,), F32), dq, dkv.cols_slice(0, H),
                       dkv.cols_slice(H, 2 * H), B, self.nh, Lq, Lk, self.d, kmask=kmask, causal=False,
                       rpr_k=rk, rpr_v=rv,
                       drpr_k=self.gb(p + "rpr_keys/embeddings") if self.rpr else None,
                       drpr_v=self.gb(p + "rpr_values/embeddings") if self.rpr else None,
                       max_rel=hp.max_relative_position, drop_p=hp.attention_dropout, sid=sid0,
                       defer_tables=(self._pending_rpr, tag) if self._defer_rpr() else None, oproj=oproj)
            self._linear_bwd(x_in, dq, p + "q_map", dx=dx_out, residual=ds,
                             ln_next=self._ln_next if fuse_tmask is None else None)
        # memory side: the gradients of all decoder layers add up in d_mem.  Default: every layer only records its
        # (dK, W_k) / (dV, W_v) pair and ONE K-segmented GEMM sums them after the decoder (see _finish_mem_grad);
        # otherwise each layer accumulates in place.
        if self._use_kseg():
            self._linear_bwd(mem, dkv.cols_slice(0, H), p + "k_map")
            self._linear_bwd(mem, dkv.cols_slice(H, 2 * H), p + "v_map")
            self._mem_segs += [(dkv.cols_slice(0, H), self.W(p + "k_map/W_0_0")),
                               (dkv.cols_slice(H, 2 * H), self.W(p + "v_map/W_0_0"))]
        else:
            self._linear_bwd(mem, dkv.cols_slice(0, H), p + "k_map", dx=d_mem, residual=d_mem)
            self._linear_bwd(mem, dkv.cols_slice(H, 2 * H), p + "v_map", dx=d_mem, residual=d_mem)
        if fuse_tmask is not None:
            # query side of the shared v_map: transpose of the averaging, then dgrad into dx and
            # wgrad / bias-grad ADDED to what the memory side wrote
            dvq = e.mat("g.%s.dvq" % tag, T, H)
            e.cumavg_bwd(datt, fuse_tmask, dvq, B, Lq, H)
            self._linear_bwd(x_in, dvq, p + "v_map", dx=dx_out, residual=dx_out, accumulate=True)
        return dx_out

    def _aan_bwd(self, dx, B, L, scope, tag, tmask, sid0, side, dx_out):
        e, hp, H = self.eng, self.hp, self.H
        T = dx.rows
        ds, dg = self._ln_bwd(dx, scope, tag, None, hp.residual_dropout, sid0 + 1, side)
        cat = e.mat(tag + ".cat", T, 2 * H)
        z = e.mat(tag + ".z", T, 2 * H)
        dz = e.mat("g.%s.dz" % tag, T, 2 * H)
        dxg = e.mat("g.%s.dxg" % tag, T, H)
        dyg = e.mat("g.%s.dyg" % tag, T, H)
        e.aan_gate_bwd(dg, z, cat, dz, dxg, dyg, T, H)
        dcat = e.mat("g.%s.dcat" % tag, T, 2 * H)
        self._linear_bwd(cat, dz, scope + "/z_project", dx=dcat)
        flags = 1 if hp.aan_mask else 0
        if hp.use_ffn:
            # through the FFN that sits between the average and the gate
            dyp = e.mat("g.%s.dyp" % tag, T, H)
            e.lib.call("zk_add_bf16", dyp.ptr, H, dyg.ptr, dyg.ld, dcat.ptr + H * 2, 2 * H, T, H, e.stream)
            h = e.mat(tag + ".h", T, self.F)
            dh = e.mat("g.%s.dh" % tag, T, self.F)
            rp = hp.relu_dropout
            self._linear_bwd(h, dyp, scope + "/ffn_layer/output", dx=dh, act=2, aux=h,
                             aux_scale=1.0 / (1.0 - rp) if rp > 0 else 1.0)
            dya = e.mat("g.%s.dya" % tag, T, H)
            self._linear_bwd(e.mat(tag + ".ya", T, H), dh, scope + "/ffn_layer/enlarge", dx=dya)
            dyg, flags = dya, flags | 2
        e.lib.call("zk_aan_bwd", dcat.ptr, dxg.ptr, dyg.ptr, ds.ptr, tmask.data_ptr(), dx_out.ptr, B, L, H, flags,
                   e.stream)
        return dx_out

    # ------------------------------------------------------------------ whole model
    def upload(self, source, target=None, trim=True, suffix=""):
        """Host ids -> device int32 (after remove_invalid_seq) through pinned staging slots (asynchronous copies on the
        current stream), then ONE launch (zk_batch_prep) for everything that depends on the ids alone: source mask,
        target mask + loss weights and -- with a target -- the grouping of the token rows by embedding id that the
        atomics-free embedding gradient reads (rounds 1-3 sorted on the host, two numpy sorts + ~10 blocking copies per
        batch; the reference's TensorFlow does it on the device, main.py:28).  Returns dict of static device buffers +
        dims.  suffix: write a second, STAGING set of the same buffers (Trainer.step prepares the next batch on a side
        stream while the previous step still reads the static set; commit() moves it over)."""
        e = self.eng
        src = np.asarray(source.cpu() if torch.is_tensor(source) else source)
        if trim:
            src = trim_columns(src)
        B, Ls = src.shape
        ids_s = e.buf("ids.src" + suffix, (B, Ls), torch.int32)
        e.h2d(ids_s, src)
        out = {"B": B, "Ls": Ls, "src": ids_s, "suffix": suffix,
               "max_id": max(self.hp.src_vocab.size(), self.hp.tgt_vocab.size())}
        ids_t, Lt = None, 0
        if target is not None:
            tgt = np.asarray(target.cpu() if torch.is_tensor(target) else target)
            if trim:
                tgt = trim_columns(tgt)
            Lt = tgt.shape[1]
            ids_t = e.buf("ids.tgt" + suffix, (B, Lt), torch.int32)
            e.h2d(ids_t, tgt)
            out.update({"Lt": Lt, "tgt": ids_t, "src_sort": self._sort_buffers("src" + suffix, B * Ls),
                        "tgt_sort": self._sort_buffers("tgt" + suffix, B * Lt)})
        if B > 0:
            out["smask"] = e.buf("smask" + suffix, (B, Ls), F32)
            if ids_t is not None:
                out["tmask"], out["tw"] = e.buf("tmask" + suffix, (B, Lt), F32), e.buf("tw" + suffix, (B, Lt), F32)
                out["tw_scale"] = float(self.hp.loss_scale)
            e.batch_prep(out)
        return out

    def commit(self, staged, extra=(), launch=True):
        """Move a batch that upload(..., suffix=...) prepared in staging buffers into the static buffers the captured
        step reads: one launch (zk_copy_many) on the current stream.  extra: more (dst, src) pairs for the same launch
        (the step's host scalars).  Returns the batch dict over the static buffers.  launch=False: no launch, the (dst,
        src) pairs are returned beside the dict (Trainer.step rewrites the copy node of the step's graph with them)."""
        e = self.eng
        B, Ls, Lt = staged["B"], staged["Ls"], staged.get("Lt", 0)
        out = {"B": B, "Ls": Ls, "suffix": ""}
        pairs = []

        def take(key, name, shape, dt):
            dst = e.buf(name, shape, dt)
            pairs.append((dst, staged[key]))
            out[key] = dst
        take("src", "ids.src", (B, Ls), torch.int32)
        if "tgt" in staged:
            out["Lt"] = Lt
            take("tgt", "ids.tgt", (B, Lt), torch.int32)
            for side, T in (("src", B * Ls), ("tgt", B * Lt)):
                d = self._sort_buffers(side, T)
                for k in ("rows", "seg", "uid", "n"):
                    pairs.append((d[k], staged[side + "_sort"][k]))
                out[side + "_sort"] = d
        if "smask" in staged:
            take("smask", "smask", (B, Ls), F32)
        if "tmask" in staged:
            take("tmask", "tmask", (B, Lt), F32)
            take("tw", "tw", (B, Lt), F32)
            out["tw_scale"] = staged["tw_scale"]
        pairs = pairs + list(extra)
        if not launch:
            return out, pairs
        e.copy_many(pairs)
        return out

    def _sort_buffers(self, name, T):
        """Device arrays zk_batch_prep fills for zk_embed_bwd_sorted: token rows grouped by embedding id (`rows`), group
        boundaries (`seg`), the id of each group (`uid`), their number (`n`, a device int)."""
        e = self.eng
        return {"rows": e.buf("sort.%s.rows" % name, (T,), torch.int32),
                "seg": e.buf("sort.%s.seg" % name, (T + 1,), torch.int32),
                "uid": e.buf("sort.%s.uid" % name, (T,), torch.int32),
                "n": e.buf("sort.%s.n" % name, (1,), torch.int32), "max_uniq": T}

    def lookup_tables(self):
        """[(variable, 'src_sort' | 'tgt_sort')]: embedding tables whose ONLY use is the lookup of one side's ids, so
        that their gradient has rows for the ids of the batch and zeros elsewhere (the reference hands such gradients
        around as tf.IndexedSlices, utils/parallel.py:142-181).  A table shared with the softmax (dense gradient) or
        looked up by both sides is not listed."""
        out = []
        if self.src_emb != self.soft_emb and self.src_emb != self.tgt_emb:
            out.append((self.src_emb, "src_sort"))
        if self.tgt_emb != self.soft_emb and self.tgt_emb != self.src_emb:
            out.append((self.tgt_emb, "tgt_sort"))
        return out

    def encode(self, batch, train, save):
        """transformer.py:15-84."""
        e, hp, H = self.eng, self.hp, self.H
        B, Ls = batch["B"], batch["Ls"]
        Ts = B * Ls
        smask = batch.get("smask")
        if smask is None:
            smask = e.buf("smask", (B, Ls), F32)
            e.make_mask(batch["src"], smask, Ts)
        x = e.mat("enc.x0", Ts, H)
        if not self.__dict__.get("_embeds_done"):      # (forward(): both embeddings went out as one launch)
            e.embed_fwd(batch["src"], self.store.s(self.src_emb), self.b("bias"), x, B, Ls, H,
                        drop_p=hp.dropout if train else 0.0, sid=9001)
        def layers(x=x):
            for l in range(hp.num_encoder_layer):
                pre = "encoder/layer_%d" % l
                x = self._self_attn_fwd(x, B, Ls, pre + "/self_attention", "e%d.sa" % l, smask, False, save,
                                        100 * l + 1, train)
                x = self._ffn_fwd(x, pre + "/feed_forward", "e%d.ff" % l, save, 100 * l + 11, train,
                                  last=(l == hp.num_encoder_layer - 1))
            return x
        # the layer stack is sentence-local all the way down: one persistent launch (zk_layer.hip) when every op of
        # it can be recorded (plain dot-product attention on the MFMA tiles), ordinary launches otherwise
        x = e.run_program(B, layers) if (not self.rpr and not self.use_side) else layers()
        return x, smask

    def decode_train(self, batch, enc, smask, train, save):
        """transformer.py:87-181 (training path: shifted inputs, causal self-attention)."""
        e, hp, H = self.eng, self.hp, self.H
        B, Ls, Lt = batch["B"], batch["Ls"], batch["Lt"]
        Tt = B * Lt
        want = float(hp.loss_scale) if train else 1.0
        if batch.get("tw_scale") == want and "tmask" in batch:
            tmask, w = batch["tmask"], batch["tw"]           # made by upload(), once per batch
        else:
            tmask = e.buf("tmask", (B, Lt), F32)
            w = e.buf("tw", (B, Lt), F32)
            e.target_stats(batch["tgt"], tmask, w, B, Lt, want)
        x = e.mat("dec.x0", Tt, H)
        if not self.__dict__.get("_embeds_done"):
            e.embed_fwd(batch["tgt"], self.store.s(self.tgt_emb), self.b("bias"), x, B, Lt, H, shift=True,
                        drop_p=hp.dropout if train else 0.0, sid=9002)
        NE = hp.num_encoder_layer
        group_kv = self.group_wgrad and e.gemm_impl == 0
        if group_kv:
            self._cross_kv_grouped(enc, hp.num_decoder_layer)
        for l in range(hp.num_decoder_layer):
            pre = "decoder/layer_%d" % l
            sid = 100 * (NE + l)
            if self.aan:
                x = self._aan_fwd(x, B, Lt, pre + "/average_attention", "d%d.aa" % l, tmask, save, sid + 1, train)
            elif not self.fuse:
                x = self._self_attn_fwd(x, B, Lt, pre + "/self_attention", "d%d.sa" % l, None, True, save,
                                        sid + 1, train)
            if group_kv and l == 0:
                self._join_side()
            x = self._cross_attn_fwd(x, enc, B, Lt, Ls, pre + "/" + self.cross, "d%d.ca" % l, smask, save,
                                     sid + 11, train, kv_ready=group_kv, fuse_tmask=tmask if self.fuse else None)
            x = self._ffn_fwd(x, pre + "/feed_forward", "d%d.ff" % l, save, sid + 21, train,
                              last=(l == hp.num_decoder_layer - 1))
        return x, tmask, w

    def loss_head(self, batch, feat, w, label_smooth, need_grad):
        """transformer.py:182-216: logits GEMM + k_ce_fused, or with ZERO_HIP_FUSED_CE=1 the fused form (the
        [T, V] fp32 logits are never written; the backward recomputes them tile by tile)."""
        e = self.eng
        B, Lt = batch["B"], batch["Lt"]
        Tt = B * Lt
        E = self.W(self.soft_emb)
        ce = e.buf("ce", (Tt,), F32)
        self._fused_ce = self.fused_ce and e.gemm_impl != 1 and self.H % 8 == 0
        if self._fused_ce:
            logits, dlogits = None, None
            lse = e.buf("lse", (Tt,), F32)
            e.logits_ce_fwd(feat, E, batch["tgt"], ce, lse, Tt, self.V, label_smooth)
            self._ce_ctx = (lse, w, label_smooth) if need_grad else None
        else:
            logits = e.mat("logits", Tt, self.Vpad, F32)
            if self.logits_tile256 and e.gemm_impl == 0 and self.H % 8 == 0:
                # 256x256 tiles, fp32 tile stored straight from the accumulators (scripts/gemm_big_bench.py:
                # 223 us against 274-291 us for the 128x128 kernels on the 4096 x 32000 x 512 problem).  bf16 logits
                # were measured and are slower (profiles/r03_negative_results.txt: a bf16 32x32 MFMA tile leaves
                # as 64-byte half lines, +118 us on a GEMM that is not output-bound, for -16 us of cross entropy)
                # (round 5: the form that issues its LDS-DMA in one place per half-workgroup -- tuning key 14 -- instead of
                # spread between the MFMA groups: 163-167 us against 171-175 in scripts/gemm_big_bench.py --sched)
                e.gemm_grouped([(feat, E, logits, Tt, self.V, self.H, None)], 0, 1, tile=(256, 256, 0))
            else:
                e.gemm(feat, E, logits, Tt, self.V, self.H, 0, 1)
            dlogits = e.mat("dlogits", Tt, self.Vpad) if need_grad else None
            e.ce_fused(logits, batch["tgt"], w if need_grad else None, ce, dlogits, Tt, self.V, label_smooth)
        per_sample = e.buf("per_sample", (B,), F32)
        loss = e.buf("loss", (1,), F32)
        e.loss_reduce(ce, batch["tgt"], per_sample, loss, B, Lt)
        return loss, per_sample, logits, dlogits

    def forward(self, batch, train=False, save=False, label_smooth=None):
        ls = self.hp.label_smooth if label_smooth is None else label_smooth
        self._lazy = self._use_lazy_ln(train, save)
        self._lazy_tags = {}
        # residual + LayerNorm inside the sub-layer output GEMMs (ZERO_HIP_SYNC_LN=0: a launch of their own)
        self._sync_ln = self.sync_ln_mode and not self._lazy and self.eng.gemm_impl == 0 and self.H % 64 == 0 and \
            self.H <= 1024 and self.F % 64 == 0 and self.eng.sync_ln_usable()
        pair_embeds = self.merge_small and "tgt" in batch and not self.eng.lib.recording
        if self._sync_ln and not pair_embeds:
            self.eng.ln_epoch_bump()          # (else the paired embedding launch below advances the epoch)
        if self._lazy:
            self._fold_weights()
        # round 6: both input embeddings depend on the ids alone -- one launch in front of the step instead of two
        # (ZERO_HIP_MERGE_SMALL=0: the round-5 launches)
        self._embeds_done = False
        if pair_embeds:
            e_, H_ = self.eng, self.H
            e_.embed_fwd_pair(batch["src"], self.store.s(self.src_emb), e_.mat("enc.x0", batch["B"] * batch["Ls"], H_),
                              batch["Ls"], 9001, batch["tgt"], self.store.s(self.tgt_emb),
                              e_.mat("dec.x0", batch["B"] * batch["Lt"], H_), batch["Lt"], 9002, self.b("bias"), batch["B"], H_,
                              drop_p=self.hp.dropout if train else 0.0, bump_epoch=self._sync_ln)
            self._embeds_done = True
        try:
            enc, smask = self.encode(batch, train, save)
            feat, tmask, w = self.decode_train(batch, enc, smask, train, save)
        finally:
            self._embeds_done = False
        loss, per_sample, logits, dlogits = self.loss_head(batch, feat, w, ls, save)
        self._lazy = False        # (encode() / decode_train() are also called directly by the decode path)
        self._sync_ln = False
        self._ctx = (batch, enc, smask, feat, tmask, dlogits)
        return loss, per_sample, logits

    def backward(self, on_ready=None):
        """Hand-written mirror of forward(train=True, save=True): fills store.grad.
        ``on_ready(key)`` is called as soon as every gradient under ``key`` (a variable
        name or ``encoder/layer_i`` / ``decoder/layer_i``) is final, so that the
        all-reduce of that bucket can overlap the rest of the backward."""
        e, hp, H = self.eng, self.hp, self.H
        if on_ready is None:
            on_ready = lambda key: None
        batch, enc, smask, feat, tmask, dlogits = self._ctx
        B, Ls, Lt = batch["B"], batch["Ls"], batch["Lt"]
        Ts, Tt = B * Ls, B * Lt
        self.fused_info = None
        if dlogits is None:      # fused cross entropy: recompute the logits tiles, write d(loss)/d(logits)
            lse, w, ls = self._ce_ctx
            dlogits = e.mat("dlogits", Tt, self.Vpad)
            e.logits_ce_bwd(feat, self.W(self.soft_emb), batch["tgt"], w, lse, dlogits, Tt, self.V, ls)
        st = self.store
        # embedding tables receive dense gradients (zero rows for unseen ids): untouched rows must
        # read as zero.  The softmax table is fully overwritten by its wgrad GEMM instead.
        if self.src_emb != self.soft_emb:
            e.zero(st.g(self.src_emb))
        if self.tgt_emb != self.soft_emb and self.tgt_emb != self.src_emb:
            e.zero(st.g(self.tgt_emb))
        # (relative-position tables: zk_attn_bwd's callers overwrite or zero-then-accumulate them, func.Engine.attn_bwd)
        # logits / softmax embedding
        E = self.W(self.soft_emb)
        P = [e.mat("gd.p0", Tt, H), e.mat("gd.p1", Tt, H)]
        cur = 0
        e.gemm(dlogits, E, P[cur], Tt, H, self.Vpad, 0, 0)
        gE = self.gW(self.soft_emb)
        if self.group_wgrad and e.gemm_impl == 0:
            # (the table's gradient is complete here only if no embedding lookup adds rows to it afterwards)
            self._pending_wgrads.append((dlogits, feat, gE, self.Vpad, H, Tt, None, None, None,
                                         self.soft_emb != self.tgt_emb and self.soft_emb != self.src_emb))
        else:
            self._side(lambda: e.gemm(dlogits, feat, gE, self.Vpad, H, Tt, 1, 0))
        d_enc = e.mat("g.denc", Ts, H)
        if not self._use_kseg():
            e.zero(d_enc.t)
        NE = hp.num_encoder_layer

        def layer_input(side, l, kind):
            # output of the sub-layer preceding `kind` in layer l == its residual input
            if side == "d":
                order = ([] if self.fuse else ["aa"] if self.aan else ["sa"]) + ["ca", "ff"]
            else:
                order = ["sa", "ff"]
            i = order.index(kind)
            if i > 0:
                return e.mat("%s%d.%s.o" % (side, l, order[i - 1]), Tt if side == "d" else Ts, H)
            if l > 0:
                return e.mat("%s%d.ff.o" % (side, l - 1), Tt if side == "d" else Ts, H)
            return e.mat("dec.x0" if side == "d" else "enc.x0", Tt if side == "d" else Ts, H)

        fuse_ln_bwd = self.sync_ln_mode and self.sync_ln_bwd and self.group_wgrad and e.gemm_impl == 0 and \
            H % 64 == 0 and H <= 1024 and not self.fuse and not self._lazy_tags and not e.lib.recording and \
            e.sync_ln_usable()       # (the fallback of a failed self-test covers the backward too)

        def ln_below(side, l, kind):
            """(scope, tag, bias of the linear layer before it, dropout, dropout site) of the LayerNorm whose output is the
            residual input of sub-layer `kind` of layer l -- its backward runs inside the dgrad launch that ends `kind`'s
            backward (zk_gemm_ln_bwd) -- or None (the embedding below the stack, the averaging sub-layer, switched off)."""
            if not fuse_ln_bwd:
                return None
            if side == "d":
                order = (["aa"] if self.aan else ["sa"]) + ["ca", "ff"]
            else:
                order = ["sa", "ff"]
            i = order.index(kind)
            if i > 0:
                pk, pl = order[i - 1], l
            elif l > 0:
                pk, pl = "ff", l - 1
            else:
                return None
            if pk == "aa":
                return None
            pre = ("decoder" if side == "d" else "encoder") + "/layer_%d" % pl
            base = 100 * (NE + pl) if side == "d" else 100 * pl
            if pk == "sa":
                scope, sid0 = pre + "/self_attention", base + 1
                pb = scope + "/dot_attention/o_map/b_0"
            elif pk == "ca":
                scope, sid0 = pre + "/" + self.cross, base + 11
                pb = scope + "/dot_attention/o_map/b_0"
            else:
                scope, sid0 = pre + "/feed_forward", base + (21 if side == "d" else 11)
                pb = scope + "/ffn_layer/output/b_0"
            return (scope, "%s%d.%s" % (side, pl, pk), pb, hp.residual_dropout, sid0 + 1)

        ready_d = []
        for l in reversed(range(hp.num_decoder_layer)):
            pre = "decoder/layer_%d" % l
            sid = 100 * (NE + l)
            self._ln_next = ln_below("d", l, "ff")
            self._ffn_bwd(P[cur], layer_input("d", l, "ff"), pre + "/feed_forward", "d%d.ff" % l, sid + 21, "d",
                          P[cur ^ 1])
            cur ^= 1
            self._ln_next = ln_below("d", l, "ca")
            self._cross_attn_bwd(P[cur], layer_input("d", l, "ca"), enc, d_enc, B, Lt, Ls,
                                 pre + "/" + self.cross, "d%d.ca" % l, smask, sid + 11, "d", P[cur ^ 1],
                                 fuse_tmask=tmask if self.fuse else None)
            cur ^= 1
            self._ln_next = None
            if self.aan:
                self._aan_bwd(P[cur], B, Lt, pre + "/average_attention", "d%d.aa" % l, tmask, sid + 1, "d",
                              P[cur ^ 1])
                cur ^= 1
            elif not self.fuse:
                self._ln_next = ln_below("d", l, "sa")
                self._self_attn_bwd(P[cur], layer_input("d", l, "sa"), B, Lt, pre + "/self_attention",
                                    "d%d.sa" % l, None, True, sid + 1, "d", P[cur ^ 1])
                cur ^= 1
            self._ln_next = None
            ready_d.append(pre)
            # (ZERO_HIP_GROUP_ALL=1, one rank: the decoder's weight gradients wait for the encoder's -- ONE grouped launch
            # per step, one partial last round of tiles instead of two)
            if not self.group_all and (len(ready_d) >= self.group_layers or l == 0):
                self._flush_wgrads()
                for key in ready_d:
                    self._side(lambda key=key: on_ready(key))
                ready_d = []
        dxt = P[cur]

        def tgt_embed_grads():
            # side stream is in order: the softmax wgrad that overwrote the shared table is done
            e.embed_bwd_sorted(batch["tgt_sort"], dxt, st.g(self.tgt_emb), H,
                               accumulate=(self.tgt_emb == self.soft_emb), drop_p=hp.dropout, sid=9002)
            e.colsum(dxt, st.g("bias"), skip_L=Lt, accumulate=False, drop_p=hp.dropout, sid=9002)
        def tables_ready():
            if self.soft_emb != self.src_emb:
                self._side(lambda: on_ready(self.soft_emb))
            if self.tgt_emb != self.soft_emb and self.tgt_emb != self.src_emb:
                self._side(lambda: on_ready(self.tgt_emb))
        if not self.group_all:
            self._side(tgt_embed_grads)
            tables_ready()
        if self._use_kseg():
            self._finish_mem_grad(d_enc)
        # encoder
        Q = [d_enc, e.mat("ge.p1", Ts, H)]
        cur = 0
        ready_e = []
        for l in reversed(range(NE)):
            pre = "encoder/layer_%d" % l
            other = Q[cur ^ 1] if (Q[cur ^ 1] is not d_enc) else e.mat("ge.p0", Ts, H)
            self._ln_next = ln_below("e", l, "ff")
            self._ffn_bwd(Q[cur], layer_input("e", l, "ff"), pre + "/feed_forward", "e%d.ff" % l, 100 * l + 11,
                          "e", other)
            Q[cur ^ 1] = other
            cur ^= 1
            other = Q[cur ^ 1] if (Q[cur ^ 1] is not d_enc) else e.mat("ge.p0", Ts, H)
            self._ln_next = ln_below("e", l, "sa")
            self._self_attn_bwd(Q[cur], layer_input("e", l, "sa"), B, Ls, pre + "/self_attention", "e%d.sa" % l,
                                smask, False, 100 * l + 1, "e", other)
            Q[cur ^ 1] = other
            cur ^= 1
            self._ln_next = None
            ready_e.append(pre)
            if (not self.group_all and len(ready_e) >= self.group_layers) or l == 0:
                self._flush_wgrads()
                for key in ready_d + ready_e:      # (group_all: the decoder's keys were held back with its weight gradients)
                    self._side(lambda key=key: on_ready(key))
                ready_d, ready_e = [], []
        dxs = Q[cur]
        # round 6 (one rank, different tables): the two gradient scatters as one launch, the two bias column sums as one pair
        pair = self.group_all and self.merge_small and self.src_emb != self.tgt_emb
        if self.group_all:
            if ready_d:                            # a model without encoder layers
                self._flush_wgrads()
                for key in ready_d:
                    self._side(lambda key=key: on_ready(key))
            if not pair:
                self._side(tgt_embed_grads)   # after the (single) grouped launch that overwrote the shared softmax table
                tables_ready()

        def src_embed_grads():
            e.embed_bwd_sorted(batch["src_sort"], dxs, st.g(self.src_emb), H,
                               accumulate=(self.src_emb == self.tgt_emb), drop_p=hp.dropout, sid=9001)
            e.colsum(dxs, st.g("bias"), skip_L=0, accumulate=True, drop_p=hp.dropout, sid=9001)
            on_ready("bias")
            on_ready(self.src_emb)

        def both_embed_grads():
            e.embed_bwd_sorted_pair(batch["tgt_sort"], dxt, st.g(self.tgt_emb), self.tgt_emb == self.soft_emb, 9002,
                                    batch["src_sort"], dxs, st.g(self.src_emb), False, 9001, H, drop_p=hp.dropout)
            e.colsum_pair(dxt, Lt, 9002, dxs, 0, 9001, st.g("bias"), drop_p=hp.dropout)
        if pair:
            self._side(both_embed_grads)
            tables_ready()
            self._side(lambda: (on_ready("bias"), on_ready(self.src_emb)))
        else:
            self._side(src_embed_grads)
        self._join_side()
