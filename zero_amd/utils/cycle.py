# coding: utf-8
"""Gradient accumulation, global-norm clipping and the Adam update on flat HBM buffers.

Counterpart of the reference's utils/cycle.py:47-135 (``create_train_op``) plus the
``tf.train.AdamOptimizer`` it wraps (main.py:178-181):

  * ``zero_op``    -> :meth:`TrainOp.zero`     (cycle.py:58-71 zero the slots)
  * ``collect_op`` -> :meth:`TrainOp.collect`  (cycle.py:73-85 slot += grad, count += 1)
  * ``train_op``   -> :meth:`TrainOp.apply`    g = (g + slot)/(count+1) (cycle.py:86-88);
    gnorm = ||g||, pnorm = ||theta|| (cycle.py:94-95); clip iff ``clip_grad_norm`` is a
    non-zero float (cycle.py:98-101); TF1 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t),
    theta -= lr_t * m / (sqrt(v) + eps).

Everything is one pass per quantity over the scope's flat buffers (zero_amd/variables.py);
the 1/N of the tower average (utils/parallel.py:184-196) and 1/loss_scale (main.py:28-30)
are folded into the same pass as ``grad_scale``.  EMA (cycle.py:113-127) is out of scope.
"""

import math

import torch

from zero_amd import hip


class TrainOp(object):
    def __init__(self, store, params, engine):
        self.store, self.hp, self.eng = store, params, engine
        dev = store.device
        # [0:6] host scalars, 6 gnorm, 7 skipped (this update), 8 ema decay, 9 gnorm bound, 10 sticky count of
        # skipped / non-finite updates (only ever incremented on the device; see bad_updates())
        self.hyper = torch.zeros(12, dtype=torch.float32, device=dev)
        if getattr(params, "safe_nan", False) and getattr(params, "gnorm_upper_bound", 0.) > 0.:
            self.hyper[9] = float(params.gnorm_upper_bound)      # main.py:325-329: update skipped above it
        # per-step host scalars travel through pinned staging slots (a single pinned buffer rewritten every step raced
        # with its own asynchronous copy once the host ran a few steps ahead of the device)
        from zero_amd.utils.queuer import PinnedRing
        self._pins = PinnedRing(dev)
        self.ema = None          # tf.train.ExponentialMovingAverage shadows (cycle.py:113-127), on demand
        self._backup = None
        if getattr(params, "ema_decay", -1.) > 0.:
            self.ema = store.master.clone()      # shadows start at the variables' initial values
        self.pnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.count = 0
        _lib = hip.lib()
        self._ws = torch.empty(max(_lib.query("zk_norm_workspace") * 2, _lib.query("zk_adam_step_workspace"),
                                   _lib.query("zk_adam_range_workspace") if _lib.experiments else 0),
                               dtype=torch.uint8, device=dev)

    # cycle.py:58-71
    def zero(self):
        if self.hp.update_cycle > 1:
            if self.store.accum is None:
                self.store.accum = torch.zeros_like(self.store.grad)
            else:
                self.eng.zero(self.store.accum)
        self.count = 0

    # cycle.py:73-85
    def collect(self):
        self.collect_launch()
        self.count += 1

    def collect_launch(self):
        """Device side of collect_op (slot += g); graph-capturable, the counter is the caller's."""
        st = self.store
        if st.accum is None:
            st.accum = torch.zeros_like(st.grad)
        self.eng.lib.call("zk_axpby_f32", st.accum.data_ptr(), st.grad.data_ptr(), 1.0, 1.0, st.numel,
                          self.eng.stream)

    def add_slots_launch(self):
        """g <- g + slot of the final micro step (cycle.py:86-88); graph-capturable."""
        st = self.store
        self.eng.lib.call("zk_axpby_f32", st.grad.data_ptr(), st.accum.data_ptr(), 1.0, 1.0, st.numel,
                          self.eng.stream)

    def set_hyper(self, lr, world=1, dst=None, seq=None):
        """Host-side scalars of this update (lr is fed per step: main.py:157,292).  dst: a staging copy of ``hyper``
        to fill instead (Trainer.step uploads the next step's scalars on its side stream; hyper_pairs() lists what the
        commit launch then copies)."""
        hyper = self.hyper if dst is None else dst
        hp = self.hp
        t = self.store.step + 1
        lr_t = lr * math.sqrt(1.0 - hp.beta2 ** t) / (1.0 - hp.beta1 ** t)
        clip = hp.clip_grad_norm or None
        clip = float(clip) if isinstance(clip, float) else 0.0
        scale = 1.0 / (float(world) * float(hp.loss_scale) * float(self.count + 1))
        import numpy as _np
        if seq is not None:
            # a staging copy (dst): one upload of all twelve words -- [6], [7], [9], [10] are the device's own in the live
            # array and never leave a staging copy; [11] carries the caller's sequence number (int32 bits), which the
            # commit launch copies to a pinned host word (Trainer.step)
            v = _np.zeros(12, dtype=_np.float32)
            v[:6] = [lr_t, hp.beta1, hp.beta2, hp.epsilon, scale, clip]
            if self.ema is not None:
                t_ = self.store.step + 1
                v[8] = min(float(hp.ema_decay), (1.0 + t_) / (10.0 + t_))
            v[11:12] = _np.array([seq], dtype=_np.int32).view(_np.float32)
            self._pins.put(hyper[:12], v)
            return scale
        self._pins.put(hyper[:6], _np.array([lr_t, hp.beta1, hp.beta2, hp.epsilon, scale, clip], dtype=_np.float32))
        if self.ema is not None:
            # num_updates = global_step after this update (ema.apply runs under train_op's control
            # dependency, cycle.py:116-118): d = min(decay, (1 + n) / (10 + n))
            self._pins.put(hyper[8:9], _np.array([min(float(hp.ema_decay), (1.0 + t) / (10.0 + t))],
                                                      dtype=_np.float32))
        return scale

    def hyper_pairs(self, staged):
        """(dst, src) copies that move staged host scalars into ``hyper`` ([6], [7], [9], [10] are the device's own)."""
        pairs = [(self.hyper[:6], staged[:6])]
        if self.ema is not None:
            pairs.append((self.hyper[8:9], staged[8:9]))
        return pairs

    def fused_ctx(self):
        """What the weight-gradient launch needs to update weights itself (zk_gemm_grouped_update): the flat buffers and
        the device scalars of THIS update -- set_hyper() must have run before the backward is issued."""
        st = self.store
        return {"master": st.master, "m": st.m, "v": st.v, "shadow": st.shadow, "grad": st.grad, "hyper": self.hyper}

    def _segments(self, ranges):
        """Device tables (seg_lo, prefix, nseg, total) of what the fused launch did NOT update: the complement of
        `ranges` in [0, numel), in elements (every variable starts on a 64-element boundary)."""
        cache = self.__dict__.setdefault("_seg_cache", {})
        ent = cache.get(ranges)
        if ent is None:
            lo, segs = 0, []
            for a, b in ranges:
                if a > lo:
                    segs.append((lo, a - lo))
                lo = max(lo, b)
            if self.store.numel > lo:
                segs.append((lo, self.store.numel - lo))
            assert all(x % 4 == 0 and n % 4 == 0 for x, n in segs)
            prefix = [0]
            for _, n in segs:
                prefix.append(prefix[-1] + n // 4)
            dev = self.store.device
            ent = (torch.tensor([x // 4 for x, _ in segs] or [0], dtype=torch.int64, device=dev),
                   torch.tensor(prefix, dtype=torch.int64, device=dev), len(segs), prefix[-1] * 4)
            cache[ranges] = ent
        return ent

    def launch_update(self, scale, advance_seed=True, fused=None):
        """Device side of train_op; graph-capturable (reads scalars from self.hyper).  Also advances the
        dropout step seed (same launch).  When the update does not depend on the global norm (no clipping,
        no safe_nan -- cycle.py:98-101 with the recipe's clip_grad_norm = 0.0) the gradient norm is only
        reported, and it is accumulated inside the Adam pass instead of a pass of its own.
        fused = (ranges, sq, n_extra) from TransformerCore.fused_info: those ranges of the flat buffers were updated
        inside the weight-gradient launch; this call updates the rest and finishes the norms over both parts."""
        st, lib, s = self.store, self.eng.lib, self.eng.stream
        if fused is not None and fused[0]:
            ranges, sq, n_extra = fused
            seg_lo, prefix, nseg, total = self._segments(ranges)
            if nseg:
                lib.call("zk_adam_step_segments", st.master.data_ptr(), st.grad.data_ptr(), st.m.data_ptr(), st.v.data_ptr(),
                         st.shadow.data_ptr(), seg_lo.data_ptr(), prefix.data_ptr(), nseg, total, self.hyper.data_ptr(),
                         self.pnorm.data_ptr(), self.eng.seed.data_ptr() if advance_seed else None, sq.data_ptr(),
                         n_extra, self._ws.data_ptr(), self._ws.numel(), s)
                if self.ema is not None:
                    lib.call("zk_ema", self.ema.data_ptr(), st.master.data_ptr(), self.hyper.data_ptr(), st.numel, s)
                return
        norm_free = self.can_update_by_range()
        if not norm_free:
            lib.call("zk_l2norm", st.grad.data_ptr(), st.numel, scale, self.hyper.data_ptr() + 6 * 4,
                     self._ws.data_ptr(), self._ws.numel(), s)
        # parameter norm (cycle.py:95) is accumulated inside the Adam pass over the same data
        lib.call("zk_adam_step", st.master.data_ptr(), st.grad.data_ptr(), st.m.data_ptr(), st.v.data_ptr(),
                 st.shadow.data_ptr(), st.numel, self.hyper.data_ptr(), self.pnorm.data_ptr(),
                 self.eng.seed.data_ptr() if advance_seed else None, 1 if norm_free else 0,
                 self.eng.sync_ln_err_ptr(),      # a step whose in-launch exchange gave up is skipped ON the device
                 self._ws.data_ptr(), self._ws.numel(), s)
        if self.ema is not None:
            lib.call("zk_ema", self.ema.data_ptr(), st.master.data_ptr(), self.hyper.data_ptr(), st.numel, s)

    # -- data parallelism: the update of a gradient bucket as soon as ITS all-reduce is done ----------
    def can_update_by_range(self):
        """Per-bucket updates need an update that does not depend on the global gradient norm: no
        clipping (cycle.py:98-101 with clip_grad_norm = 0.0) and no safe_nan skip."""
        hp = self.hp
        clip = hp.clip_grad_norm or None
        return not isinstance(clip, float) and not getattr(hp, "safe_nan", False)

    # -- the norm-free update in pieces, norms included (single rank: parts of it overlap the backward) -----------
    def launch_update_slot(self, lo, hi, slot):
        """TF1 Adam on elements [lo, hi) (64-element aligned) + its partial sums of squares into workspace slot."""
        st, lib, s = self.store, self.eng.lib, self.eng.stream
        lib.call("zk_adam_range", st.master.data_ptr() + lo * 4, st.grad.data_ptr() + lo * 4, st.m.data_ptr() + lo * 4,
                 st.v.data_ptr() + lo * 4, st.shadow.data_ptr() + lo * 2, hi - lo, self.hyper.data_ptr(), slot,
                 self._ws.data_ptr(), self._ws.numel(), s)

    def finish_update_slots(self, nslots):
        """Gradient / parameter norms from the slots, flags, seed advance, EMA."""
        st, lib, s = self.store, self.eng.lib, self.eng.stream
        lib.call("zk_adam_finish", self.hyper.data_ptr(), self.pnorm.data_ptr(), self.eng.seed.data_ptr(), nslots,
                 self._ws.data_ptr(), self._ws.numel(), s)
        if self.ema is not None:
            lib.call("zk_ema", self.ema.data_ptr(), st.master.data_ptr(), self.hyper.data_ptr(), st.numel, s)

    def begin_update_by_range(self):
        """hyper[6] (the norm the Adam kernel guards on) is only known after the last bucket: a finite
        placeholder lets the per-bucket updates through; finish_update_by_range() writes the real value."""
        self.hyper[6:7].fill_(1.0)

    def launch_update_range(self, lo, hi):
        """TF1 Adam on elements [lo, hi) of the flat buffers (bucket boundaries are 64-element aligned)."""
        st, lib, s = self.store, self.eng.lib, self.eng.stream
        n = hi - lo
        lib.call("zk_adam", st.master.data_ptr() + lo * 4, st.grad.data_ptr() + lo * 4, st.m.data_ptr() + lo * 4,
                 st.v.data_ptr() + lo * 4, st.shadow.data_ptr() + lo * 2, n, self.hyper.data_ptr(), None, None, 0, s)

    def finish_update_by_range(self, scale):
        """Global gradient norm (logging + the NaN report of main.py:316-319; the update has been applied,
        as in the reference without safe_nan), parameter norm, EMA."""
        st, lib, s = self.store, self.eng.lib, self.eng.stream
        nb = self._ws.numel() // 2
        lib.call("zk_l2norm", st.grad.data_ptr(), st.numel, scale, self.hyper.data_ptr() + 6 * 4,
                 self._ws.data_ptr(), nb, s)
        lib.call("zk_norm_flag", self.hyper.data_ptr(), s)      # non-finite norm -> flag + sticky count
        lib.call("zk_l2norm", st.master.data_ptr(), st.numel, 1.0, self.pnorm.data_ptr(), self._ws.data_ptr() + nb, nb, s)
        if self.ema is not None:
            lib.call("zk_ema", self.ema.data_ptr(), st.master.data_ptr(), self.hyper.data_ptr(), st.numel, s)

    # cycle.py:120-127: evaluate with the averaged weights, then put the raw ones back
    def ema_backup(self):
        if self.ema is not None:
            self._backup = self.store.master.clone()

    def ema_assign(self):
        if self.ema is not None:
            self.store.master.copy_(self.ema)
            self.store.refresh_shadow()

    def ema_restore(self):
        if self.ema is not None and self._backup is not None:
            self.store.master.copy_(self._backup)
            self.store.refresh_shadow()
            self._backup = None

    def apply(self, lr, world=1, launch=True):
        st = self.store
        if self.count > 0:   # cycle.py:86-88: g <- g + slot, scaled by 1/(count+1) below
            self.eng.lib.call("zk_axpby_f32", st.grad.data_ptr(), st.accum.data_ptr(), 1.0, 1.0, st.numel,
                              self.eng.stream)
        scale = self.set_hyper(lr, world)
        if launch:
            self.launch_update(scale)
        st.step += 1
        self.count = 0
        return scale

    def stats(self):
        """(gradient_norm, parameter_norm, skipped) of the LAST update -- forces a sync; call at display time."""
        h = self.hyper.cpu()
        g = float(h[6])
        return g, float(self.pnorm.cpu()[0]), bool(h[7] != 0) or not math.isfinite(g)

    def bad_updates(self):
        """Number of updates so far that were skipped or saw a non-finite gradient norm (sticky device counter:
        nothing between two reads is lost).  Forces a sync; the loop reads it before every display and every
        checkpoint (main.py:316-319 checks every step, which here would serialise host and device)."""
        return int(self.hyper[10:11].cpu()[0])
