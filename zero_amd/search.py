# coding: utf-8
"""Beam search over the HIP decode step.

Same algorithm and the same public function as the reference's search.py:19-275
(``beam_search(features, encoding_fn, decoding_fn, params)`` -> {'seq', 'score'}):
alive / finished beam sets, initial log-probs [0, f32.min, ...] so that step 0 expands only
beam 0 (search.py:46), EOS forbidden at step 0 (search.py:152-155), GNMT length penalty
((5+t+1)/6)^alpha (search.py:168-170), 2K candidates per sentence, alive top-K of the
unfinished ones and finished top-K of (previous K + current 2K) (search.py:198-228), stop
when every sentence's worst finished score beats the best possible alive score or the
length cap is hit (search.py:85-113).  "Greedy" decoding = ``beam_size=1``.

Division of labour per step: the model step, log-softmax, penalty and the top-2K over
K*V candidates run on the GPU (one fused kernel, zk_beam_topk); the O(B*K) alive/finished
bookkeeping on the 2K survivors runs on the host in fp32 numpy exactly as search.py writes
it (it needs the termination test on the host anyway); caches are reordered on the GPU.
"""

import os

import numpy as np
import torch

from zero_amd.utils import dtype as zdtype

F32_MIN = np.finfo(np.float32).min


def _top_k(x, k):
    """tf.nn.top_k: descending, ties -> lower index."""
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, axis=-1), idx


def beam_search(features, encoding_fn, decoding_fn, params):
    """search.py:19-275.  Runs on the engine's work stream (decode-step graphs cannot be captured on
    the legacy default stream)."""
    if params.search_mode == "cache":
        from zero_amd.models._factory import get_core
        from zero_amd.models._decode import STARTUP_LOCK
        # a lane's first batch builds its core (device allocations), its work stream and an event here: under the
        # start-up lock, like every other allocation of a batch's start-up -- another lane may be inside a stream
        # capture, which an allocation in this thread would invalidate (models/_decode.py)
        with STARTUP_LOCK:
            eng = get_core(params, params.model_name).eng
            cur = torch.cuda.current_stream(eng.device)
            ws = eng.work_stream
            ws.wait_stream(cur)
        with torch.cuda.stream(ws):
            out = _beam_search(features, encoding_fn, decoding_fn, params)
        cur.wait_stream(ws)
        return out
    return _beam_search(features, encoding_fn, decoding_fn, params)


def _beam_search(features, encoding_fn, decoding_fn, params):
    box = []
    try:
        return _beam_search_body(features, encoding_fn, decoding_fn, params, box)
    finally:
        # the batch's start-up lock (models/_decode.py: allocations and graph captures of concurrent batches are
        # serialised) must not outlive the search, whatever happened
        if box and hasattr(box[0], "get"):
            from zero_amd.models._decode import startup_end
            startup_end(box[0])


def _beam_search_body(features, encoding_fn, decoding_fn, params, box):
    f32 = np.float32
    K = params.beam_size
    alpha = params.decode_alpha
    eos_id = params.tgt_vocab.eos()
    pad_id = params.tgt_vocab.pad()
    V = params.tgt_vocab.size()
    source = features["source"]
    src_np = np.asarray(source.cpu() if torch.is_tensor(source) else source)
    B = src_np.shape[0]
    src_len = (src_np != 0).sum(-1).astype(f32)
    max_target_length = src_len + f32(params.decode_length)
    cache_mode = params.search_mode == "cache"
    if cache_mode:
        # start-up (allocations, pinned staging, the two graph captures) is serialised between concurrent batches
        # (models/_decode.py STARTUP_LOCK); step_static releases it once both step graphs exist, _beam_search at the end
        from zero_amd.models import _decode as _dec
        _dec.startup_begin()
        try:
            state = encoding_fn(source, beam_size=K,
                                max_steps=int(max_target_length.max()) + 2)
        except BaseException:
            _dec.STARTUP_LOCK.release()
            raise
        state["_startup_held"] = True
        box.append(state)
        core = state["_core"]
    else:
        from zero_amd.models._factory import get_core
        core = get_core(params, params.model_name)
        state = np.repeat(src_np, K, axis=0)      # expand_tile_dims of the raw source
    e = core.eng
    dev = e.device

    log_probs = np.tile(np.array([[0.] + [F32_MIN] * (K - 1)], dtype=f32), (B, 1))
    scores = np.zeros_like(log_probs)
    seq = np.full((B, K, 1), pad_id, dtype=np.int64)
    fin_seq = np.zeros_like(seq)
    fin_scores = np.full((B, K), F32_MIN, dtype=f32)
    fin_flags = np.zeros((B, K), dtype=bool)

    d_prev = e.buf("bs.prev", (B * K,), torch.float32)
    d_ts = e.buf("bs.ts", (B, 2 * K), torch.float32)
    d_ti = e.buf("bs.ti", (B, 2 * K), torch.int32)
    d_tok = e.buf("bs.tok", (B * K,), torch.int32)
    d_idx = e.buf("bs.idx", (B * K,), torch.int32)
    mtl_i = max_target_length.astype(np.int32)
    time = 0
    # AAN decoding: the whole step (cache reorder + model + top-2K) has static launch arguments and is
    # replayed as a hipGraph; per-step scalars travel through a small device buffer
    static_step = cache_mode and state.get("static_ok", False) and hasattr(decoding_fn, "step_static") \
        and os.environ.get("ZERO_HIP_DECODE_GRAPH", "1") != "0"
    if static_step and K <= 16 and os.environ.get("ZERO_HIP_DECODE_HOST_C", "1") != "0":
        Tcap = int(state["Tmax"]) + 2
        if os.environ.get("ZERO_HIP_DECODE_DEVICE_BOOK", "1") != "0" and 2 * K * Tcap * 4 <= 64 * 1024 \
                and getattr(params, "search_trace", None) is None:
            return _beam_search_device(state, decoding_fn, params, B, K, V, eos_id, pad_id, alpha, max_target_length)
        return _beam_search_static(state, decoding_fn, params, B, K, V, eos_id, pad_id, alpha, max_target_length)
    if static_step:
        BK = B * K
        pack = state["pack_host"].numpy()             # pinned; shares memory with the tensor
        pack[2 * BK:3 * BK] = np.arange(BK, dtype=np.int32)     # step 0: identity reorder
        out_host = state["out_host"]
        out_np = out_host.numpy()
    while True:
        # ---- search.py:85-113
        max_lp = np.power((f32(5.) + max_target_length) / f32(6.), f32(alpha)).astype(f32)
        best_alive = log_probs[:, 0] / max_lp
        worst_fin = (fin_scores * fin_flags.astype(f32)).min(axis=1)
        worst_fin = worst_fin + (f32(1.) - fin_flags.any(axis=1).astype(f32)) * F32_MIN
        bound_is_met = bool((worst_fin > best_alive).all())
        length_is_met = bool((time < mtl_i).any())
        if bound_is_met or not length_is_met:
            break
        # ---- model step (search.py:118-142)
        penalty = f32(np.power(f32((f32(5.) + f32(time + 1)) / f32(6.)), f32(alpha)))
        if static_step:
            if time >= state["Tmax"]:
                raise RuntimeError("decode step %d exceeds the allocated cache length %d" % (time, state["Tmax"]))
            pack[0:BK] = seq[:, :, -1].reshape(-1)
            pack[BK:2 * BK] = log_probs.reshape(-1).view(np.int32)
            pack[3 * BK], pack[3 * BK + 1], pack[3 * BK + 2] = \
                time, int(np.float32(penalty).view(np.int32)), (eos_id if time < 1 else -1)
            state["pack_dev"].copy_(state["pack_host"], non_blocking=True)
            decoding_fn.step_static(state, params.beam_search_temperature, zdtype.inf())
            logits = None
        elif cache_mode:
            d_tok.copy_(torch.from_numpy(seq[:, :, -1].reshape(-1).astype(np.int32)))
            logits, state = decoding_fn(d_tok, state, time)
        else:
            flat = seq.reshape(B * K, -1)
            decode_target = np.concatenate([flat[:, 1:], np.ones((B * K, 1), dtype=flat.dtype)], axis=1)
            logits, state = decoding_fn(decode_target, state, time)
        # ---- fused log-softmax + penalty + top-2K (search.py:143-176)
        if not static_step:
            if params.enable_noise_beam_search:      # search.py:143-145
                e.lib.call("zk_add_gumbel", logits.ptr, B * K, V, logits.ld, float(zdtype.epsilon()),
                           e.seed.data_ptr(), 7001, e.stream)
                e.lib.call("zk_seed_advance", e.seed.data_ptr(), 1, e.stream)
            d_prev.copy_(torch.from_numpy(log_probs.reshape(-1)))
            e.beam_topk(logits, d_prev, d_ts, d_ti, B, K, V, 2 * K, params.beam_search_temperature, penalty,
                        eos_id if time < 1 else -1, zdtype.inf())
        if static_step:
            out_host.copy_(state["out_dev"])           # one D2H for scores and indices (synchronises)
            topk_scores = out_np[0].view(f32).copy()
            topk_idx = out_np[1].astype(np.int64)
        else:
            topk_scores = d_ts.cpu().numpy().astype(f32)
            topk_idx = d_ti.cpu().numpy().astype(np.int64)
        beam_idx = topk_idx // V
        sym_idx = topk_idx % V
        bpos = np.arange(B)[:, None]
        curr_seq = np.concatenate([seq[bpos, beam_idx], sym_idx[:, :, None]], axis=2)
        # ---- alive (search.py:192-210)
        curr_fin = (sym_idx == eos_id) | (time >= mtl_i)[:, None]
        alive_scores, alive_idx = _top_k(topk_scores + curr_fin.astype(f32) * F32_MIN, K)
        alive_seq = curr_seq[bpos, alive_idx]
        alive_beam = beam_idx[bpos, alive_idx]
        with np.errstate(over="ignore"):
            alive_lp = (alive_scores * penalty).astype(f32)
        # ---- finished (search.py:212-228)
        cfs = topk_scores + (f32(1.) - curr_fin.astype(f32)) * F32_MIN
        all_flags = np.concatenate([fin_flags, curr_fin], axis=1)
        all_scores = np.concatenate([fin_scores, cfs], axis=1)
        fin_scores, fin_idx = _top_k(all_scores, K)
        fin_flags = all_flags[bpos, fin_idx]
        pad_col = np.full((B, K, 1), pad_id, dtype=seq.dtype)
        all_seq = np.concatenate([np.concatenate([fin_seq, pad_col], axis=2), curr_seq], axis=1)
        fin_seq = all_seq[bpos, fin_idx]
        seq, log_probs, scores = alive_seq, alive_lp, alive_scores
        if cache_mode:
            flat_idx = (np.arange(B)[:, None] * K + alive_beam).reshape(-1).astype(np.int32)
            if static_step:
                pack[2 * BK:3 * BK] = flat_idx          # travels with the next step's inputs; it reorders first
            else:
                d_idx.copy_(torch.from_numpy(flat_idx))
                state.reorder(d_idx)
        time += 1

    if cache_mode:
        _release_graphs(state)
    any_fin = fin_flags.any(axis=1)
    final_seqs = np.where(any_fin[:, None, None], fin_seq, seq)
    final_scores = np.where(any_fin[:, None], fin_scores, scores)
    return {"seq": final_seqs[:, :, 1:], "score": final_scores, "steps": time}


def _release_graphs(state):
    """End of a batch: its step graphs go to the per-shape cache of their core or are destroyed (models/_decode.py)."""
    if hasattr(state, "get") and state.get("graphs"):
        from zero_amd.models._decode import retire_graphs
        retire_graphs(state)


def _beam_search_static(state, decoding_fn, params, B, K, V, eos_id, pad_id, alpha, max_target_length):
    """The cache-mode search with the whole device step replayed from a hipGraph and the host bookkeeping
    of search.py:85-113,168-228 done by two C calls (zk_beam_host_should_stop / zk_beam_host_step) on
    int32 [B, K, Tcap] sequence buffers.  Same arithmetic as the numpy path of _beam_search (kept for the
    eager / dev modes; the tests hold the two against the same oracle)."""
    import ctypes
    f32 = np.float32
    core = state["_core"]
    lib = core.eng.lib
    BK = B * K
    Tcap = int(state["Tmax"]) + 2
    seq = np.full((B, K, Tcap), pad_id, dtype=np.int32)
    fin_seq = np.zeros((B, K, Tcap), dtype=np.int32)
    log_probs = np.tile(np.array([[0.] + [F32_MIN] * (K - 1)], dtype=f32), (B, 1))
    scores = np.zeros((B, K), dtype=f32)
    fin_scores = np.full((B, K), F32_MIN, dtype=f32)
    fin_flags = np.zeros((B, K), dtype=np.uint8)
    mtl = np.ascontiguousarray(max_target_length, dtype=f32)
    mtl_i = mtl.astype(np.int32)
    pack = state["pack_host"].numpy()             # pinned: [tok | prev log-probs | reorder index | step scalars]
    pack[0:BK] = pad_id                            # BOS = pad id (search.py:50)
    pack[2 * BK:3 * BK] = np.arange(BK, dtype=np.int32)
    out_host = state["out_host"]
    out_np = out_host.numpy()
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    tok, prev, idx = pack[0:BK], pack[BK:2 * BK], pack[2 * BK:3 * BK]
    p_seq, p_fin, p_lp, p_sc, p_fs, p_ff = P(seq), P(fin_seq), P(log_probs), P(scores), P(fin_scores), P(fin_flags)
    p_mtl, p_mtli, p_ts, p_ti, p_idx, p_tok = P(mtl), P(mtl_i), P(out_np[0]), P(out_np[1]), P(idx), P(tok)
    stop = lib.raw("zk_beam_host_should_stop")
    step = lib.raw("zk_beam_host_step")
    # params.search_trace (a list, diagnostic): the 2K (score, flat index) survivors of every step, as they came off
    # the device -- lets a test find the step at which this search first leaves another one's path
    trace = getattr(params, "search_trace", None)
    time = 0
    while True:
        if stop(B, K, p_lp, p_fs, p_ff, p_mtl, p_mtli, time, float(alpha)):
            break
        if time >= state["Tmax"]:
            raise RuntimeError("decode step %d exceeds the allocated cache length %d" % (time, state["Tmax"]))
        penalty = f32(np.power(f32((f32(5.) + f32(time + 1)) / f32(6.)), f32(alpha)))
        prev[:] = log_probs.reshape(-1).view(np.int32)
        pack[3 * BK], pack[3 * BK + 1], pack[3 * BK + 2] = \
            time, int(penalty.view(np.int32)), (eos_id if time < 1 else -1)
        state["pack_dev"].copy_(state["pack_host"], non_blocking=True)
        decoding_fn.step_static(state, params.beam_search_temperature, zdtype.inf())
        out_host.copy_(state["out_dev"])           # one D2H for scores and indices (synchronises)
        if trace is not None:
            trace.append((out_np[0].view(f32).copy(), out_np[1].copy()))
        rc = step(B, K, V, Tcap, time, p_ts, p_ti, p_seq, p_fin, p_lp, p_sc, p_fs, p_ff, p_mtli, eos_id, pad_id,
                  float(penalty), p_idx, p_tok)
        if rc != 0:
            raise RuntimeError("zk_beam_host_step failed (rc=%d)" % rc)
        time += 1
    _release_graphs(state)
    any_fin = fin_flags.any(axis=1)
    n = time + 1
    final_seqs = np.where(any_fin[:, None, None], fin_seq[:, :, :n], seq[:, :, :n]).astype(np.int64)
    final_scores = np.where(any_fin[:, None], fin_scores, scores)
    return {"seq": final_seqs[:, :, 1:], "score": final_scores, "steps": time}


def _beam_search_device(state, decoding_fn, params, B, K, V, eos_id, pad_id, alpha, max_target_length):
    """The cache-mode search with the bookkeeping of search.py:85-113,168-228 RESIDENT ON THE DEVICE
    (zk_beam_dev_prepare / zk_beam_dev_advance are the first and last node of the step graph): a decode step
    needs no host round trip; the host replays the graph ZERO_HIP_DECODE_POLL (default 1) times between reads of
    the stop flag, one group behind the launches.  Once the stop test fires the state is frozen, so the replays
    past it change nothing and the result is the one of _beam_search_static (tests hold the two equal, bit for
    bit)."""
    f32 = np.float32
    core = state["_core"]
    e = core.eng
    BK = B * K
    Tcap = int(state["Tmax"]) + 2
    mtl = np.ascontiguousarray(max_target_length, dtype=f32)
    # one int32 arena: [ctrl 4 | pen_table Tcap | max_lp B | mtl_i B | log_probs | scores | fin_scores |
    #                   fin_flags (BK each) | seq | fin_seq (BK*Tcap each)]
    sizes = [4, Tcap, B, B, BK, BK, BK, BK, BK * Tcap, BK * Tcap]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    total = int(offs[-1])
    pinned = core.__dict__.get("_book_host")          # pinned staging, kept across batches (pinning costs ~0.2 ms)
    if pinned is None or pinned[0].numel() < total or len(pinned) < 4:
        pinned = (torch.zeros(total + total // 4, dtype=torch.int32).pin_memory(),
                  torch.zeros(4, dtype=torch.int32).pin_memory(), torch.zeros(4, dtype=torch.int32).pin_memory(),
                  torch.zeros(8, dtype=torch.int32).pin_memory())       # [3]: the two control slots of zk_beam_dev_run
        core._book_host = pinned
    host, ctrl_host = pinned[0][:total], pinned[1]
    host.zero_()
    ctrl_host.zero_()
    h = host.numpy()
    part = lambda i: h[offs[i]:offs[i + 1]]
    part(1).view(f32)[:] = [f32(np.power(f32((f32(5.) + f32(t + 1)) / f32(6.)), f32(alpha))) for t in range(Tcap)]
    part(2).view(f32)[:] = [f32(np.power(f32((f32(5.) + m) / f32(6.)), f32(alpha))) for m in mtl]
    part(3)[:] = mtl.astype(np.int32)
    part(4).view(f32)[:] = np.tile(np.array([0.] + [F32_MIN] * (K - 1), dtype=f32), B)
    part(6).view(f32)[:] = F32_MIN
    part(8)[:] = pad_id                               # seq; fin_seq starts at zeros (search.py:52-57)
    dev = e.buf("bs.book", (total,), torch.int32)
    dev.copy_(host, non_blocking=True)
    d = lambda i: dev[offs[i]:offs[i + 1]].data_ptr()
    pack = state["pack_host"].numpy()
    pack[0:BK] = pad_id                               # BOS = pad id (search.py:50)
    pack[BK:2 * BK] = part(4)                         # previous log-probs of step 0
    pack[2 * BK:3 * BK] = np.arange(BK, dtype=np.int32)
    state["pack_dev"].copy_(state["pack_host"], non_blocking=True)
    state["book"] = (d(0), state["stepbuf"].data_ptr(), d(1), d(2), d(3), state["ts"].data_ptr(), state["ti"].data_ptr(),
                     d(8), d(9), d(4), d(5), d(6), d(7), state["idx"].data_ptr(), state["tok"].data_ptr(),
                     state["prev"].data_ptr(), B, K, V, Tcap, int(state["Tmax"]), eos_id, pad_id)
    # The stop flag is read one group of replays BEHIND the launches: while the host waits for the copy issued
    # after group g, group g+1 is already queued, so the device never idles on the poll.  At most 2*poll-1
    # replays run past the stop (on a frozen state).
    poll = max(1, int(os.environ.get("ZERO_HIP_DECODE_POLL", "1")))
    ctrl_dev = dev[0:4]
    stream = torch.cuda.current_stream(e.device)
    slots = [(ctrl_host[0:4], torch.cuda.Event()), (pinned[2], torch.cuda.Event())]
    launched, group, pending = 0, 0, None
    # Once both parity graphs of the step exist (at once when the shape's graphs were adopted from the cache), the rest of
    # the loop runs inside ONE C call (zk_beam_dev_run: same launches, same one-group-behind poll): no interpreter between
    # two decode steps, and with several batches in flight the lanes' host threads stop serialising on it.
    run_c = True
    from zero_amd.models import _decode as _dec
    if run_c and "_gkey" not in state:
        _dec.adopt_graphs(state, state["book"], params.beam_search_temperature, zdtype.inf(),
                          params.enable_noise_beam_search)
    stopped = False
    while True:
        if run_c and _dec._startup_settled(state):
            if pending is not None:            # drain the Python loop's poll before handing over
                stream.synchronize()
                ctrl_host = slots[(group - 1) & 1][0]
                if int(ctrl_host[1]) or launched > Tcap + 2 * poll:
                    stopped = True
                    break
            _dec.startup_end(state)
            import ctypes
            p8 = pinned[3]
            p8.zero_()
            g = state["graphs"]
            n_c, newest = ctypes.c_int(0), ctypes.c_int(0)
            e.lib.call("zk_beam_dev_run", g[0], g[1], int(state["_pp"]), ctrl_dev.data_ptr(), p8.data_ptr(),
                       int(Tcap + 2 * poll - launched), poll, stream.cuda_stream, ctypes.byref(n_c), ctypes.byref(newest))
            launched += n_c.value
            if n_c.value & 1:                  # the python-side ping-pong pointers follow the replays
                state["_pp"] = 1 - state["_pp"]
                state.bind_caches()
            ctrl_host = p8[4 * newest.value:4 * newest.value + 4]
            stopped = True
            break
        for _ in range(poll):
            decoding_fn.step_static(state, params.beam_search_temperature, zdtype.inf())
        launched += poll
        buf, ev = slots[group & 1]
        buf.copy_(ctrl_dev, non_blocking=True)
        ev.record(stream)
        if pending is not None:
            pending[1].synchronize()
            ctrl_host = pending[0]
            if int(ctrl_host[1]) or launched > Tcap + 2 * poll:
                break
        pending = (buf, ev)
        group += 1
    stream.synchronize()
    if not stopped:
        ctrl_host = slots[group & 1][0]               # the newest copy: same stop state, final step count
    _release_graphs(state)
    state.pop("book", None)
    if int(ctrl_host[2]) or not int(ctrl_host[1]):
        raise RuntimeError("decode step %d exceeds the allocated cache length %d" % (int(ctrl_host[0]), state["Tmax"]))
    host.copy_(dev)
    time = int(h[0])
    n = time + 1
    seq = part(8).reshape(B, K, Tcap)
    fin_seq = part(9).reshape(B, K, Tcap)
    scores, fin_scores = part(5).view(f32).reshape(B, K), part(6).view(f32).reshape(B, K)
    fin_flags = part(7).reshape(B, K) != 0
    any_fin = fin_flags.any(axis=1)
    final_seqs = np.where(any_fin[:, None, None], fin_seq[:, :, :n], seq[:, :, :n]).astype(np.int64)
    final_scores = np.where(any_fin[:, None], fin_scores, scores)
    return {"seq": final_seqs[:, :, 1:], "score": final_scores, "steps": time}


def decode_hypothesis(seqs, params):
    """evalu.py:14-46: take beam 0, cut at the first eos or pad."""
    out = []
    for b in range(seqs.shape[0]):
        ids = []
        for t in seqs[b, 0]:
            t = int(t)
            if t == params.tgt_vocab.eos() or t == params.tgt_vocab.pad():
                break
            ids.append(t)
        out.append(ids)
    return out
