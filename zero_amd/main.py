# coding: utf-8
"""The step section of the reference's main.py on the HIP path.

Counterparts:
  * ``tower_train_graph`` (main.py:22-45)  -> :func:`tower_train_graph`: one tower per
    process; ``train_fn`` -> loss + gradients; the tower average is the RCCL all-reduce of
    zero_amd/utils/parallel.py, launched bucket by bucket during the backward.
  * the hot loop of ``train`` (main.py:268-332) -> :class:`Trainer`: zero_op at cycle start,
    collect_op for the first update_cycle-1 micro steps, train_op on the last, LR fed as a
    host scalar evaluated before the step (main.py:280; lrs/noamlr.py).
  * ``tower_infer_graph`` (main.py:48-62) / ``tower_score_graph`` (main.py:118-130)
    -> :func:`tower_infer_graph` / :func:`tower_score_graph`.

Data loading, evaluation cadence, checkpoints and logging of main.py are host control plane
outside the hot path.
"""

import os
import time

import torch

from zero_amd import lrs
from zero_amd.models import model as model_registry
from zero_amd.models import load_all
from zero_amd.utils import parallel
from zero_amd.utils.cycle import TrainOp
from zero_amd.hip import ZeroHipError


def tower_train_graph(train_features, graph, params, reducer=None):
    """main.py:22-45 for this rank's tower.  Returns (loss tensor, train_fn output)."""
    out = graph.train_fn(train_features, params, on_ready=reducer.ready if reducer is not None else None)
    return out["loss"], out


def tower_score_graph(eval_features, graph, params):
    """main.py:118-130."""
    return graph.score_fn(eval_features, params)["score"]


def tower_infer_graph(eval_features, graph, params):
    """main.py:48-62."""
    from zero_amd.search import beam_search
    encoding_fn, decoding_fn = graph.infer_fn(params)
    out = beam_search(eval_features, encoding_fn, decoding_fn, params)
    return out["seq"], out["score"]


def _os_environ_int(name):
    import os as _os
    try:
        return int(_os.environ.get(name, "0"))
    except ValueError:
        return 0


class Trainer(object):
    """One data-parallel replica of the training step (main.py:255-332)."""

    def __init__(self, params, initializer=None):
        load_all()
        self.params = params
        self.graph = model_registry.get_model(params.model_name)
        from zero_amd.models._factory import get_core
        self.core = get_core(params, params.model_name, initializer)
        self.store = self.core.store
        self.train_op = TrainOp(self.store, params, self.core.eng)
        self.reducer = parallel.GradientAllReduce(self.store)
        self.lr = lrs.get_lr(params)
        self.global_step = 0
        self.cycle_counter = 0
        self._graphs = {}
        # test hook: run the multi-rank (segmented) capture path with a single rank
        import os as _os
        self.force_segmented = _os.environ.get("ZERO_HIP_FORCE_SEGMENTED", "0") != "0"
        # data parallelism: update each gradient bucket behind its own all-reduce (see _reduced_update)
        self.overlap_update = _os.environ.get("ZERO_HIP_OVERLAP_UPDATE", "1") != "0"
        # pad both sides of a batch to a multiple of this many positions (1 = the reference's exact shapes); see prepare_static
        self.pad_len = max(1, int(_os.environ.get("ZERO_HIP_PAD_LEN", "1")))
        # one rank: update the decoder-side parameters on a side stream while the encoder backward still runs
        # (the Adam pass is HBM-bound, the backward chain latency-bound: they overlap well); needs the norm-free
        # update (no clipping / safe_nan), see _train_and_update
        # measured on MI355X (bench.py, same box): 5.01 ms with the overlap, 4.94 ms without -- the streaming Adam
        # blocks take HBM bandwidth and wave slots from the latency-bound chain they run beside, and the fork / join
        # edges of the graph cost the rest -- so it is opt-in
        # (an experiment: needs a `make EXPERIMENTS=1` library for zk_adam_range / zk_adam_finish)
        self.overlap_adam = _os.environ.get("ZERO_HIP_OVERLAP_ADAM", "0") != "0" and self.core.eng.lib.experiments
        self._adam_stream = None
        self._empty_ids = None
        # the update of the weight matrices inside the weight-gradient launch (zk_gemm_grouped_update; _can_fuse_update):
        # MEASURED SLOWER (profiles/r04_negative_results.txt: the launch 500 -> 794 us for 255 us less in the Adam pass)
        # -> an experiment: `make EXPERIMENTS=1` library + ZERO_HIP_FUSED_UPDATE=1
        self.fuse_update = _os.environ.get("ZERO_HIP_FUSED_UPDATE", "0") == "1" and self.core.eng.lib.experiments
        self.reseed()

    # -- row-sparse exchange of lookup-table gradients (utils/parallel.py:142-181) -----------------
    def _sparse_capacity(self):
        """Slots of a row payload: the most token rows one batch of one side can have under the batching limits
        (data.py token_indexer: count * max_len < token_size unless a single sample is longer; batch mode:
        batch_size sentences of at most max_len + 1 positions).  Derived from the hyper-parameters alone, hence the
        same on every rank."""
        hp = self.params
        over = _os_environ_int("ZERO_HIP_SPARSE_ROWS")
        if over:
            return over
        if getattr(hp, "batch_or_token", "token") == "batch":
            return int(hp.batch_size) * (int(hp.max_len) + 1)
        return max(int(hp.token_size), int(hp.max_len) + 1)

    def _declare_sparse(self, batch):
        """Tell the reducer which tables of THIS step's batch travel as (ids, rows) payloads.  batch None = a tower
        without sentences: it still takes part in the all-gather, with zero rows."""
        red = self.reducer
        red.clear_sparse()
        hp = self.params
        if parallel.world_size() == 1 or not red.sparse_enabled or self.pad_len > 1 or hp.update_cycle > 1:
            return
        cap = self._sparse_capacity()
        for key, sort_name in self.core.lookup_tables():
            if batch is None:
                if self._empty_ids is None:
                    self._empty_ids = torch.zeros(4, dtype=torch.int32, device=self.store.device)
                uid, n, rows = self._empty_ids, self._empty_ids[:1], 0
            else:
                srt = batch[sort_name]
                uid, n, rows = srt["uid"], srt["n"], srt["max_uniq"]
            red.set_sparse(key, self.store.g(key), uid, n, cap, rows=rows)

    def reseed(self):
        """Position the dropout / Gumbel-noise stream: high word = params.random_seed mixed with the rank (every
        data-parallel replica draws its own masks, as every tower of the reference has its own dropout ops), low
        word = number of micro steps taken so far, so a resumed run continues the stream instead of replaying
        it from step 0.  The device advances the low word by one per micro step."""
        hp = self.params
        hi = (int(getattr(hp, "random_seed", 1234)) * 0x9E3779B1 + parallel.rank() * 0x85EBCA77 + 0x165667B1) & 0x7FFFFFFF
        micro = (self.global_step * max(1, int(hp.update_cycle)) + self.cycle_counter) & 0xFFFFFFFF
        self.core.eng.set_seed((hi << 32) | micro)

    def _can_fuse_update(self):
        """One rank, no accumulation, an update that does not look at the global norm (cycle.py:98-101 with the recipe's
        clip_grad_norm = 0.0 and no safe_nan), all weight gradients in the backward's single deferred launch."""
        return self.fuse_update and parallel.world_size() == 1 and not self.force_segmented and \
            self.params.update_cycle == 1 and self.train_op.can_update_by_range() and self.core.group_all and \
            not self.overlap_adam and not self.core.use_side

    def rollback_skipped_update(self):
        """main.py:320-332 (safe_nan): a skipped update does not run train_op, so global_step, Adam's t and the
        learning-rate schedule do not advance."""
        self.global_step -= 1
        self.store.step -= 1

    # -- eager path (any shapes) --------------------------------------------------
    def micro_step(self, features):
        """One forward+backward (+ update on the last micro step of a cycle).
        Returns the loss tensor (device, not synchronised)."""
        hp = self.params
        world = parallel.world_size()
        if self.cycle_counter == 0:
            self.train_op.zero()
        last = (self.cycle_counter + 1) >= hp.update_cycle
        self.lr.step(self.global_step)
        overlap = last and hp.update_cycle == 1
        if overlap and world > 1:
            # upload here (train_fn accepts the uploaded batch) so that the reducer knows the touched ids of the
            # lookup tables before the backward reports them
            if "B" not in features and features["source"].shape[0] > 0:
                features = self.core.upload(features["source"], features["target"])
            self._declare_sparse(features if "B" in features else None)
        fuse = last and hp.update_cycle == 1 and self._can_fuse_update()
        if fuse:
            # the weight-gradient launch runs the update of the weights itself: the scalars of this update first
            self.train_op.count = 0
            scale = self.train_op.set_hyper(self.lr.get_lr(), world)
            self.core.fused_update = self.train_op.fused_ctx()
        self.core.fused_info = None
        try:
            loss, _ = tower_train_graph(features, self.graph, hp, self.reducer if overlap else None)
        finally:
            self.core.fused_update = None
        if fuse:
            self.train_op.launch_update(scale, fused=self.core.fused_info)
            self.store.step += 1
            self.cycle_counter = 0
            self.global_step += 1
            return loss
        if not last:
            self.train_op.collect()
            self.cycle_counter += 1
            # a fresh dropout stream position for every micro batch of the cycle
            self.core.eng.lib.call("zk_seed_advance", self.core.eng.seed.data_ptr(), 1, self.core.eng.stream)
            return loss
        if hp.update_cycle > 1:
            # reference semantics: one average over N*c micro batches (cycle.py:86-88)
            scale = self.train_op.apply(self.lr.get_lr(), world, launch=False)
            self.reducer.all_reduce_everything()
            self.train_op.launch_update(scale)
        else:
            self.reducer.wait()
            self.train_op.apply(self.lr.get_lr(), world)
        self.cycle_counter = 0
        self.global_step += 1          # (the update launch advanced the dropout seed)
        return loss

    def on_work_stream(self):
        """``with trainer.on_work_stream(): ...`` -- the calling loop issues its steps from the engine's work stream (step
        graphs cannot be captured or replayed on the legacy default stream), so that step() needs no stream hand-off."""
        return torch.cuda.stream(self.core.eng.work_stream)

    # staging sets of the rotating-batch loop (see step())
    STAGE_SLOTS = max(2, int(os.environ.get("ZERO_HIP_STAGE_SLOTS", "16")))

    # -- one entry point for the training loop: captured steps whenever the shapes allow ---------
    def step(self, features, use_graph=True):
        """One micro step on ``features`` (update on the last one of a cycle), replaying a captured
        hipGraph per batch shape.  Same arithmetic and bookkeeping as :meth:`micro_step`."""
        hp = self.params
        if not use_graph or features["source"].shape[0] == 0 or self.core.use_side:
            return self.micro_step(features)
        if hp.update_cycle == 1 and self.pad_len == 1:
            # The batch is uploaded (asynchronous copies through pinned slots) and prepared (zk_batch_prep: masks, loss
            # weights, token rows grouped by embedding id) on a SIDE stream into one of STAGE_SLOTS staging sets -- in the
            # steady state, where the host runs ahead of the device, that happens while the PREVIOUS step is still running
            # -- and the captured step then starts with one small copy launch (commit) instead of waiting for a ~60-us
            # single-workgroup sort.  The step itself is unchanged; one graph per batch shape as before.
            #
            # A staging set may be overwritten once the commit that read it is done.  Rounds 4-5 had two sets and an
            # event recorded on the work stream behind every commit, for the upload stream to wait on: that event costs
            # 40-65 us PER STEP of kernel time inside the step (in-kernel stamps, scripts/step_stamps.py: with it the
            # step's interior is that much longer than a static replay's, without it the two are equal) -- wherever it
            # is recorded, with or without its system fence, and still 40 us at one record per eight steps.  So no event
            # is recorded on the work stream at all: the commit launch itself copies the step's sequence number (it
            # travels with the host scalars) into a PINNED HOST word, and the HOST does not enqueue the upload into set
            # n % S before that word says commit n - S + 1 has been reached (kernels of a stream run in order: commit
            # n - S is complete then).  In the steady state the host is a few steps ahead of the device and never waits.
            eng = self.core.eng
            cur = torch.cuda.current_stream(eng.device)
            ws, up = eng.work_stream, eng.upload_stream
            S = self.STAGE_SLOTS
            n = self._stage_n = getattr(self, "_stage_n", -1) + 1
            slot = n % S
            done = self.__dict__.get("_commit_reached")
            if done is None:
                done = self._commit_reached = torch.full((1,), -1, dtype=torch.int32).pin_memory()
            if n >= S:
                spins = 0
                while int(done[0]) < n - S + 1:          # plain host read of the pinned word (no HIP call)
                    spins += 1
                    time.sleep(0 if spins < 50 else 2e-4)
                    if spins > 300000:                   # ~a minute: the device is not making progress
                        raise ZeroHipError("Trainer.step: commit %d was never reached (pinned word %d)" % (n - S + 1, int(done[0])))
            self.lr.step(self.global_step)
            self.train_op.count = 0
            with torch.cuda.stream(up):
                staged = self.core.upload(features["source"], features["target"], suffix=".stg%d" % slot)
                # the step's host scalars (lr_t, ...) and its sequence number travel the same way: no copy of their own
                # between two step graphs
                hstage = eng.buf("hyper.stg%d" % slot, (12,), torch.float32)
                scale = self.train_op.set_hyper(self.lr.get_lr(), parallel.world_size(), dst=hstage, seq=n)
                ups = self.__dict__.setdefault("_upload_events", {})
                ev_up = ups.get(slot)
                if ev_up is None:
                    ev_up = ups[slot] = torch.cuda.Event()
                ev_up.record(up)
            # (a caller that already runs on the engine's work stream -- zero_amd.main.train and bench.py do, via
            # Trainer.on_work_stream() -- pays no cross-stream hand-off per step: two event packets fewer between two
            # step graphs, profiles/r05_rocprof_captured_step_gaps*.txt)
            same = cur == ws
            if not same:
                ws.wait_stream(cur)
            ws.wait_event(ev_up)
            with torch.cuda.stream(ws):
                loss = self._step_staged(staged, hstage, scale)
            if not same:
                cur.wait_stream(ws)
            return loss
        if hp.update_cycle == 1:
            eng = self.core.eng
            cur = torch.cuda.current_stream(eng.device)
            ws = eng.work_stream
            ws.wait_stream(cur)
            with torch.cuda.stream(ws):
                self.prepare_static(features)
                loss = self._step_static(True)
            cur.wait_stream(ws)
            return loss
        eng = self.core.eng
        cur = torch.cuda.current_stream(eng.device)
        ws = eng.work_stream
        ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            loss = self._step_accumulate(features)
        cur.wait_stream(ws)
        return loss

    def _step_staged(self, staged, hstage, scale):
        """The step on a batch that sits prepared in a staging set (Trainer.step, on the work stream).

        Single rank, whole-step graph: the commit launch (staging set -> the buffers the step reads, zk_copy_many) is the
        FIRST NODE of the step's graph -- one submission per step, as for a static replay -- and its arguments are
        rewritten in the instantiated graph before every launch (zk_graph_set_copy_many; the graphs with that node are
        keyed ("stg", B, Ls, Lt), apart from the ones step_static() captures for the same shape).  Everything else (first
        sight of a shape, several ranks, the side-stream variants): the commit launch in front, then _step_static().
        Either way the commit also copies the step's sequence number into the pinned word step() polls."""
        eng = self.core.eng
        extra = self.train_op.hyper_pairs(hstage) + [(self._commit_reached.view(torch.float32), hstage[11:12])]
        fast = parallel.world_size() == 1 and not self.force_segmented and not self.core.use_side and \
            os.environ.get("ZERO_HIP_COMMIT_IN_GRAPH", "1") != "0"
        key = ("stg", staged["B"], staged["Ls"], staged.get("Lt", 0))
        self._check_graph_cache()
        probe = None
        if fast:
            # the in-graph commit needs the whole commit in ONE zk_copy_many launch (its graph must hold exactly one such
            # node, zk_graph_set_copy_many): a pair more than the launch takes (today: 13 batch pairs + the hyper / EMA /
            # sequence words = 16 = the limit) falls back to the commit-in-front route instead of raising
            probe = self.core.commit(staged, extra=extra, launch=False)
            fast = eng.copy_many_fits(probe[1])
        g = self._graphs.pop(key, None) if fast else None
        if g is None:
            if fast:
                self._graphs[key] = "warm"                 # eager now (sizes every scratch buffer), captured next time
            self.batch = self.core.commit(staged, extra=extra)
            self._declare_sparse(self.batch)
            return self._step_static(not fast, scale=scale)     # fast: eager this once; else the usual routes
        if g == "warm":
            def body():
                self.batch = self.core.commit(staged, extra=extra)
                self._train_and_update(scale)
            g = eng.graph_capture(body)
        else:
            self.batch, pairs = probe
            if not eng.graph_set_copy_many(g, pairs):
                raise ZeroHipError("Trainer.step: the commit of a batch no longer fits one zk_copy_many launch")
        self._graphs[key] = g                              # most recently used last
        self._declare_sparse(self.batch)
        eng.graph_launch(g)
        self.store.step += 1
        self.global_step += 1
        return eng.buf("loss", (1,), torch.float32)

    def _graph_run(self, key, body):
        eng = self.core.eng
        g = self._graphs.pop(key, None)
        if g is None:                       # first sight of a shape: eager, sizes the scratch buffers
            self._graphs[key] = "warm"
            body()
            return
        if g == "warm":
            g = eng.graph_capture(body)
        self._graphs[key] = g
        eng.graph_launch(g)

    def _step_accumulate(self, features):
        """update_cycle > 1 (cycle.py:73-92): 'collect' micro steps then the final one."""
        hp, eng = self.params, self.core.eng
        world = parallel.world_size()
        self._check_graph_cache()
        if self.cycle_counter == 0:
            self.train_op.zero()
        last = (self.cycle_counter + 1) >= hp.update_cycle
        self.lr.step(self.global_step)
        batch = self.prepare_static(features)
        shape = (batch["B"], batch["Ls"], batch["Lt"])
        loss = eng.buf("loss", (1,), torch.float32)
        if not last:
            self._graph_run(("acc",) + shape,
                            lambda: (self.graph.train_fn(batch, hp), self.train_op.collect_launch(),
                                     eng.lib.call("zk_seed_advance", eng.seed.data_ptr(), 1, eng.stream)))
            self.train_op.count += 1
            self.cycle_counter += 1
            return loss
        scale = self.train_op.set_hyper(self.lr.get_lr(), world)

        def tail():
            self.train_op.launch_update(scale)
        if world == 1:
            self._graph_run(("fin",) + shape,
                            lambda: (self.graph.train_fn(batch, hp), self.train_op.add_slots_launch(), tail()))
        else:       # one exchange per update over the accumulated gradient (cycle.py:86-88)
            self._graph_run(("finA",) + shape,
                            lambda: (self.graph.train_fn(batch, hp), self.train_op.add_slots_launch()))
            self.reducer.all_reduce_everything()
            self._graph_run(("finB",), tail)
        self.train_op.count = 0
        self.store.step += 1
        self.cycle_counter = 0
        self.global_step += 1
        return loss

    # -- data-parallel captured path: hipGraph segments between the gradient-bucket hand-offs ----
    def _step_segmented(self, scale):
        """Multi-rank step with the launch cost of the single-rank one.  RCCL calls stay outside
        hipGraphs, so the launch sequence is captured as SEGMENTS: every time the backward reports
        a finished gradient bucket (``on_ready``) the running capture is closed and a new one is
        opened; replay = launch segment, hand the finished buckets to the all-reduce (which then
        overlaps the next segment), ..., wait, launch the captured norm + Adam update."""
        import ctypes
        hp, eng = self.params, self.core.eng
        lib = eng.lib
        key = ("seg", self.batch["B"], self.batch["Ls"], self.batch["Lt"])
        plan = self._graphs.get(key)

        def eager():
            self.graph.train_fn(self.batch, hp, on_ready=self.reducer.ready)
            self._reduced_update(scale, None)
        if plan is None or getattr(self, "_seg_disabled", False):
            # first use of a shape: eager (sizes every scratch buffer); also the fallback should a
            # capture ever fail on a platform (the job then keeps running, only slower)
            self._graphs.setdefault(key, "warm")
            return eager()
        if plan == "warm":
            try:
                plan = self._capture_segments(scale)
            except Exception as exc:       # leave capture mode cleanly, run this and later steps eagerly
                import logging
                logging.getLogger("zero_amd").warning("segmented hipGraph capture failed (%s); eager steps", exc)
                try:
                    ex = ctypes.c_void_p()
                    lib.call("zk_graph_end", torch.cuda.current_stream(eng.device).cuda_stream, ctypes.byref(ex))
                except Exception:
                    pass
                self._seg_disabled = True
                return eager()
            self._graphs[key] = plan
        for kind, what in plan:
            if kind == "graph":
                eng.graph_launch(what)
            elif kind == "ready":
                for k in what:
                    self.reducer.ready(k)
            else:
                self._reduced_update(scale, what)

    def _reduced_update(self, scale, update_graph):
        """Everything after the last gradient hand-off.  When the update does not depend on the global norm
        (no clipping, no safe_nan) every bucket is updated as soon as its own all-reduce has finished, so
        the Adam pass hides behind the collectives still in flight; otherwise wait for all, then the
        (captured) norm + Adam."""
        eng = self.core.eng
        top = self.train_op
        if self.overlap_update and parallel.world_size() > 1 and top.can_update_by_range():
            top.begin_update_by_range()
            done = 0
            for lo, hi in self.reducer.drain():
                top.launch_update_range(lo, hi)
                done += hi - lo
            if done != self.store.numel:          # a variable group nobody reported: never leave it stale
                raise RuntimeError("gradient buckets covered %d of %d elements" % (done, self.store.numel))
            top.finish_update_by_range(scale)
            eng.lib.call("zk_seed_advance", eng.seed.data_ptr(), 1, eng.stream)
            return
        self.reducer.wait()
        if update_graph is not None:
            eng.graph_launch(update_graph)
        else:
            top.launch_update(scale)

    def _capture_segments(self, scale):
        import ctypes
        hp, eng = self.params, self.core.eng
        lib = eng.lib
        if True:
            stream = torch.cuda.current_stream(eng.device).cuda_stream
            plan, mark = [], [0]

            def begin():
                lib.call("zk_graph_begin", stream)
                mark[0] = lib.ncalls

            def cut():
                ex = ctypes.c_void_p()
                lib.call("zk_graph_end", stream, ctypes.byref(ex))
                return ex

            def on_ready(k):
                if lib.ncalls == mark[0] and plan and plan[-1][0] == "ready":
                    plan[-1][1].append(k)          # nothing enqueued since the last hand-off
                    return
                plan.append(("graph", cut()))
                plan.append(("ready", [k]))
                begin()

            begin()
            try:
                self.graph.train_fn(self.batch, hp, on_ready=on_ready)
            finally:
                plan.append(("graph", cut()))
            begin()
            try:
                self.train_op.launch_update(scale)
            finally:
                plan.append(("update", cut()))
            return plan

    def _train_and_update(self, scale):
        """Single rank, update_cycle == 1: forward + backward + update.  With a norm-free update (cycle.py:98-101,
        clip_grad_norm 0.0, no safe_nan) the parameters whose gradients are final -- the whole decoder side and the
        target / softmax embedding once the decoder backward is through -- are updated on a side stream beside the
        encoder backward; the rest follows on the main stream; the norms come out of the same passes
        (zk_adam_range slots -> zk_adam_finish).  Same arithmetic per element as the single launch."""
        hp, top, core = self.params, self.train_op, self.core
        if not (self.overlap_adam and top.can_update_by_range() and not core.use_side):
            core.fused_update = top.fused_ctx() if self._can_fuse_update() else None
            core.fused_info = None
            try:
                self.graph.train_fn(self.batch, hp)
            finally:
                core.fused_update = None
            top.launch_update(scale, fused=core.fused_info)
            return
        eng = core.eng
        main = torch.cuda.current_stream(eng.device)
        if self._adam_stream is None:
            self._adam_stream = torch.cuda.Stream(eng.device)
        side = self._adam_stream
        ranges = self.reducer.ranges
        early = set(k for k in ranges if k.startswith("decoder/"))
        if core.soft_emb != core.src_emb:
            early.add(core.soft_emb)
            if core.tgt_emb != core.src_emb:
                early.add(core.tgt_emb)
        state = {"slots": 0, "forked": False, "seen": set(), "pend": []}

        def launch(rs):
            rs = sorted(rs)
            merged = []
            for lo, hi in rs:
                if merged and merged[-1][1] == lo:
                    merged[-1][1] = hi
                else:
                    merged.append([lo, hi])
            for lo, hi in merged:
                top.launch_update_slot(lo, hi, state["slots"])
                state["slots"] += 1
            return sum(hi - lo for lo, hi in merged)

        covered = [0]

        def ready(key):
            if key in state["seen"]:
                return
            state["seen"].add(key)
            state["pend"].append(ranges[key])
            if not state["forked"] and early and early <= state["seen"]:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    covered[0] += launch(state["pend"])
                state["pend"] = []
                state["forked"] = True
        self.graph.train_fn(self.batch, hp, on_ready=ready)
        covered[0] += launch(state["pend"])
        if covered[0] != self.store.numel:          # a variable group nobody reported: never leave it stale
            raise RuntimeError("parameter ranges covered %d of %d elements" % (covered[0], self.store.numel))
        if state["forked"]:
            ev = torch.cuda.Event()
            ev.record(side)
            main.wait_event(ev)
        top.finish_update_slots(state["slots"])

    # -- captured path (static shapes, update_cycle == 1) ---------------------------
    def prepare_static(self, features):
        """Upload one batch into the static id buffers; later steps may overwrite the
        same buffers (same shapes) with :meth:`refill`."""
        src, tgt = features["source"], features["target"]
        m = self.pad_len
        if m > 1:
            # Length bucketing for the shape-keyed graph cache: both sides padded (id 0 = pad) up to a multiple of m.
            # Padded source keys are masked, padded target positions follow every real one (causal) and carry no loss:
            # the real tokens' values and every gradient are unchanged (with dropout > 0 the random stream is indexed by
            # the padded row numbers, i.e. another draw of the same distribution).  Token-sized batches otherwise
            # produce a new (B, Ls, Lt) -- one eager step + one capture -- almost every step.
            from zero_amd.models._core import trim_columns
            src, tgt = (trim_columns(np.asarray(t.cpu() if torch.is_tensor(t) else t)) for t in (src, tgt))
            src = np.pad(src, ((0, 0), (0, -src.shape[1] % m)))
            tgt = np.pad(tgt, ((0, 0), (0, -tgt.shape[1] % m)))
            self.batch = self.core.upload(src, tgt, trim=False)
            self._declare_sparse(self.batch)
            return self.batch
        self.batch = self.core.upload(src, tgt)
        self._declare_sparse(self.batch)
        return self.batch

    def step_static(self, use_graph=True):
        """One full update on the static batch: fwd + bwd (+ all-reduce) + Adam.
        With a single rank the whole step is one hipGraph replay."""
        eng = self.core.eng
        cur = torch.cuda.current_stream(eng.device)
        ws = eng.work_stream
        if cur == ws:                    # (Trainer.on_work_stream: no hand-off)
            return self._step_static(use_graph)
        ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            loss = self._step_static(use_graph)
        cur.wait_stream(ws)
        return loss

    MAX_GRAPHS = 96      # cached step graphs (one per batch shape); least recently used goes first

    def _check_graph_cache(self):
        """Captured graphs hold raw buffer addresses: drop them all when the engine replaced a buffer
        (a larger batch shape arrived), and keep the cache bounded."""
        eng = self.core.eng
        if getattr(self, "_graph_gen", None) != eng.realloc_gen:
            for g in self._graphs.values():
                for h in (g if isinstance(g, list) else [("graph", g)]):
                    if h[0] in ("graph", "update") and not isinstance(h[1], str):
                        eng.lib.call("zk_graph_destroy", h[1])
            self._graphs = {}
            self._graph_gen = eng.realloc_gen
        while len(self._graphs) > self.MAX_GRAPHS:
            key = next(iter(self._graphs))
            g = self._graphs.pop(key)
            for h in (g if isinstance(g, list) else [("graph", g)]):
                if h[0] in ("graph", "update") and not isinstance(h[1], str):
                    eng.lib.call("zk_graph_destroy", h[1])

    def _step_static(self, use_graph, scale=None):
        """scale: the caller has already staged this step's host scalars (Trainer.step: side stream + commit launch)."""
        hp = self.params
        assert hp.update_cycle == 1, "captured step supports update_cycle == 1"
        world = parallel.world_size()
        eng = self.core.eng
        self._check_graph_cache()
        if scale is None:
            self.lr.step(self.global_step)
            self.train_op.count = 0
            scale = self.train_op.set_hyper(self.lr.get_lr(), world)
        if world == 1 and use_graph and not self.force_segmented:
            key = (self.batch["B"], self.batch["Ls"], self.batch["Lt"])
            g = self._graphs.pop(key, None)
            if g is not None:
                self._graphs[key] = g          # most recently used last
            if g is None:
                # first use of a shape runs eagerly once (sizes every scratch buffer), then
                # the same launch sequence is captured
                self._graphs[key] = "warm"
                return self._step_static(False, scale=scale)
            if g == "warm":
                def body():
                    self._train_and_update(scale)
                g = eng.graph_capture(body)
                self._graphs[key] = g
            eng.graph_launch(g)
        elif use_graph and not self.core.use_side and (world > 1 or self.force_segmented):
            self._step_segmented(scale)
        elif world > 1:
            self.graph.train_fn(self.batch, hp, on_ready=self.reducer.ready)
            self._reduced_update(scale, None)
        else:
            self._train_and_update(scale)
        self.store.step += 1
        self.global_step += 1
        return eng.buf("loss", (1,), torch.float32)


# ---------------------------------------------------------------------------------------------
# The callers either side of the step (SURVEY.md §8(f)-1,4): main.py:133-466 (train),
# main.py:469-535 (evaluate), main.py:538-607 (scorer).  Host control flow only; every number
# comes from the HIP paths above.
# ---------------------------------------------------------------------------------------------
import logging
import os
import time

import numpy as np

log = logging.getLogger("zero_amd")


def _restore(trainer, saver, path=None):
    from zero_amd.utils.saver import assign_tensors
    tensors = saver.restore(path)
    if tensors is None:
        return False
    got, missing, step = assign_tensors(trainer.store, trainer.params.scope_name or "model", tensors)
    for name in missing:
        log.warning("%s is missed", name)
    if step is not None:
        # tf.train.Saver restores global_step and Adam's beta powers together with the slots, also from
        # ``pretrained_model`` (main.py:221-226, saver.py:131-170): warm m / v keep their bias correction
        trainer.global_step = step
        trainer.store.step = step
        trainer.reseed()
    if trainer.train_op.ema is not None:
        from zero_amd.utils.saver import assign_flat
        assign_flat(trainer.store, trainer.train_op.ema, trainer.params.scope_name or "model", tensors,
                    "/ExponentialMovingAverage")
    log.info("restored %d variables (%d missing)", len(got), len(missing))
    return True


def _evaluate_dev(trainer, dataset, params, target_file, tag):
    from zero_amd import evalu
    trainer.train_op.ema_backup()
    trainer.train_op.ema_assign()
    t0 = time.time()
    tranes, scores, indices = evalu.decoding(trainer.graph, dataset, params)
    bleu = evalu.eval_metric(tranes, target_file, indices=indices)
    trainer.train_op.ema_restore()
    log.info("GStep %s, Scores %s, BLEU %s, Duration %.3f s", tag, np.mean(scores) if scores else 0.0, bleu,
             time.time() - t0)
    if parallel.rank() == 0:      # every rank decodes the (small) dev set; one of them writes the file
        evalu.dump_tanslation(tranes, os.path.join(params.output_dir, "eval-{}.trans.txt".format(tag)),
                              indices=indices)
    return bleu, scores


def train(params):
    """main.py:133-466: epochs over the bitext, update_cycle micro steps per update, periodic
    save / dev-set BLEU / early stopping, final evaluation.  Returns the best dev score."""
    from zero_amd.data import Dataset
    from zero_amd.utils import queuer
    from zero_amd.utils.saver import Saver, collect_tensors
    rec = params.recorder
    if rec.estop or rec.epoch > params.epoches or rec.step > params.max_training_steps:
        log.info("Stop condition reached, you have finished training your model.")
        return 0.
    train_dataset = Dataset(params.src_train_file, params.tgt_train_file, params.src_vocab, params.tgt_vocab,
                            params.max_len, batch_or_token=params.batch_or_token,
                            data_leak_ratio=params.data_leak_ratio)
    dev_dataset = Dataset(params.src_dev_file, params.src_dev_file, params.src_vocab, params.src_vocab,
                          params.eval_max_len, batch_or_token='batch', data_leak_ratio=params.data_leak_ratio)
    trainer = Trainer(params)
    rank, world = parallel.rank(), parallel.world_size()
    saver = Saver(checkpoints=params.checkpoints, output_dir=params.output_dir,
                  best_checkpoints=params.best_checkpoints)
    if params.pretrained_model:
        _restore(trainer, saver, params.pretrained_model)
    _restore(trainer, saver)
    trainer.lr.lrate = rec.lrate
    scope = params.scope_name or "model"

    def checkpoint(gstep, score=None):
        if rank == 0:
            saver.save(collect_tensors(trainer.store, scope, gstep, params, ema=trainer.train_op.ema), gstep, score)
            rec.save_to_json(os.path.join(params.output_dir, "record.json"))

    start_time, cum_tokens = time.time(), 0
    bad_seen = trainer.train_op.bad_updates()
    pending = []
    # the loop issues everything from the engine's work stream (Trainer.on_work_stream): the steps need no stream
    # hand-off, and the loop's own reads (loss, norms, checkpoints, dev-set decoding) are ordered behind them on that stream
    on_ws = trainer.on_work_stream() if trainer.core.eng.device.type == "cuda" else None
    if on_ws is not None:
        on_ws.__enter__()
    try:
        best = _train_epochs(params, trainer, rec, saver, checkpoint, train_dataset, dev_dataset, log, rank, world, pending,
                             start_time, cum_tokens, bad_seen, queuer)
    finally:
        if on_ws is not None:
            on_ws.__exit__(None, None, None)
            torch.cuda.current_stream(trainer.core.eng.device).wait_stream(trainer.core.eng.work_stream)
    return best


def _train_epochs(params, trainer, rec, saver, checkpoint, train_dataset, dev_dataset, log, rank, world, pending, start_time,
                  cum_tokens, bad_seen, queuer):
    """The epoch loop of train() (main.py:255-466), issued from the engine's work stream."""
    for epoch in range(rec.epoch, params.epoches + 1):
        rec.epoch = epoch
        log.info("Training the model for epoch %d", epoch)
        size = params.batch_size if params.batch_or_token == 'batch' else params.token_size
        feed = queuer.EnQueuer(
            train_dataset.batcher(size, buffer_size=params.buffer_size, shuffle=params.shuffle_batch, train=True),
            lambda x: x, worker_processes_num=params.process_num,
            input_queue_size=params.input_queue_size, output_queue_size=params.output_queue_size)
        trainer.lr.before_epoch(eidx=epoch)
        for lidx, data in enumerate(feed):
            if params.train_continue and lidx <= rec.lidx:
                continue
            rec.lidx = lidx
            pending.append(data)          # one batch per tower (main.py:268-273)
            if len(pending) < world:
                continue
            data = pending[rank]
            pending = []
            cum_tokens += int(np.sum(data['tgt'] > 0))
            last = (trainer.cycle_counter + 1) >= params.update_cycle
            # the step of every batch SHAPE is captured once and replayed (eager on first sight,
            # captured on the second, replayed from the third on), with or without accumulation
            loss = trainer.step({"source": data['src'], "target": data['tgt']})
            if not last:
                continue
            gstep = trainer.global_step
            will_save = gstep > 0 and (gstep % params.save_freq == 0 or gstep % params.eval_freq == 0)
            if gstep % params.disp_freq == 0 or params.safe_nan or will_save:
                gnorm, pnorm, skipped = trainer.train_op.stats()
                loss_v = float(loss.reshape(-1)[0].cpu()) if hasattr(loss, "cpu") else float(loss)
                # the device counts every skipped / non-finite update (sticky): nothing that happened between two
                # reads is lost, and it is read before anything is written to disk
                bad_total = trainer.train_op.bad_updates()
                bad_new, bad_seen = bad_total > bad_seen, bad_total
                # the in-launch LayerNorm exchange (zk_gemm_add_ln ..) bounds every wait and records a give-up on the
                # device: results since the last read would be wrong -- stop loudly before anything is written
                if trainer.core.eng.sync_ln_errors():
                    from zero_amd.hip import ZeroHipError
                    raise ZeroHipError("a workgroup of an in-launch LayerNorm exchange gave up waiting for its peers "
                                       "(zk_gemm_add_ln / zk_gemm_ln_bwd / zk_attn_out_ln); rerun with ZERO_HIP_SYNC_LN=0")
                if skipped or bad_new or not np.isfinite(loss_v) or not np.isfinite(gnorm):
                    if not params.safe_nan:          # main.py:316-319
                        log.error("Nan or Inf raised! Loss %s GNorm %s.", loss_v, gnorm)
                        rec.estop = True
                        break
                    if skipped:                      # main.py:320-332: the step is passed, global_step stays
                        log.error("Nan or Inf raised, GStep %s is passed! Loss %s GNorm %s.", gstep, loss_v, gnorm)
                        trainer.rollback_skipped_update()
                        continue
                if gstep % params.disp_freq == 0:
                    now = time.time()
                    log.info("Epoch %d, GStep %d~%d, LStep %d~%d, Loss %.3f, GNorm %.3f, PNorm %.3f, Lr %.5f, "
                             "Src %s, Tgt %s, Tokens %d, UD %.3f s", epoch, gstep - params.disp_freq + 1, gstep,
                             lidx - params.disp_freq + 1, lidx, loss_v, gnorm, pnorm, trainer.lr.get_lr(),
                             data['src'].shape, data['tgt'].shape, cum_tokens, now - start_time)
                    start_time, cum_tokens = now, 0
            if gstep > 0 and gstep % params.save_freq == 0:
                checkpoint(gstep)
            if gstep > 0 and gstep % params.eval_freq == 0:
                bleu, scores = _evaluate_dev(trainer, dev_dataset, params, params.tgt_dev_file, gstep)
                checkpoint(gstep, bleu)
                valid = [v[1] for v in rec.valid_script_scores]
                if not valid or bleu > np.max(valid):
                    rec.bad_counter = 0
                else:
                    rec.bad_counter += 1
                    if rec.bad_counter > params.estop_patience:
                        rec.estop = True
                        break
                rec.history_scores.append((int(gstep), float(np.mean(scores)) if scores else 0.0))
                rec.valid_script_scores.append((int(gstep), float(bleu)))
                if rank == 0:
                    rec.save_to_json(os.path.join(params.output_dir, "record.json"))
                trainer.lr.after_eval(float(bleu))
            if gstep >= params.max_training_steps:
                rec.estop = True
                break
            rec.step = int(gstep)
            rec.lrate = trainer.lr.lrate
        if rec.estop:
            log.info("Early Stopped!")
            break
        rec.lidx = -1
        trainer.lr.after_epoch(eidx=epoch)
    log.info("Start Final Evaluating")
    _evaluate_dev(trainer, dev_dataset, params, params.tgt_dev_file, int(rec.step + 1))
    log.info("Your training is finished :)")
    return saver.best_score


def _eval_trainer(params):
    """Model + restored (and, with ema_decay > 0, averaged) weights for test / score modes."""
    from zero_amd.utils.saver import Saver
    trainer = Trainer(params)
    saver = Saver(checkpoints=params.checkpoints, output_dir=params.output_dir)
    tensors = saver.restore(params.output_dir)
    if tensors is not None:
        from zero_amd.utils.saver import assign_tensors
        scope = params.scope_name or "model"
        if params.ema_decay > 0.:
            # main.py:507-514: evaluate with the ExponentialMovingAverage shadows when present
            for key in list(tensors):
                ema_key = key + "/ExponentialMovingAverage"
                if ema_key in tensors:
                    tensors[key] = tensors[ema_key]
        assign_tensors(trainer.store, scope, tensors)
    return trainer


def evaluate(params):
    """main.py:469-535: translate the test set, BLEU against ``tgt_test_file``, dump to
    ``test_output``."""
    from zero_amd import evalu
    from zero_amd.data import Dataset
    dataset = Dataset(params.src_test_file, params.src_test_file, params.src_vocab, params.src_vocab,
                      params.eval_max_len, batch_or_token='batch', data_leak_ratio=params.data_leak_ratio)
    trainer = _eval_trainer(params)
    t0 = time.time()
    tranes, scores, indices = evalu.decoding(trainer.graph, dataset, params)
    bleu = evalu.eval_metric(tranes, params.tgt_test_file, indices=indices)
    log.info("Scores %s, BLEU %s, Duration %ss", np.mean(scores) if scores else 0.0, bleu, time.time() - t0)
    evalu.dump_tanslation(tranes, params.test_output, indices=indices)
    return bleu


def scorer(params):
    """main.py:538-607: per-sentence scores of (src_test_file, tgt_test_file) + perplexity."""
    from zero_amd import evalu
    from zero_amd.data import Dataset
    dataset = Dataset(params.src_test_file, params.tgt_test_file, params.src_vocab, params.tgt_vocab,
                      params.eval_max_len, batch_or_token='batch', data_leak_ratio=params.data_leak_ratio)
    trainer = _eval_trainer(params)
    t0 = time.time()
    scores, ppl = evalu.scoring(trainer.graph, dataset, params)
    log.info("Scores %s, PPL %s, Duration %ss", np.mean(scores), ppl, time.time() - t0)
    evalu.dump_tanslation(scores, params.test_output)
    return float(np.mean(scores))
