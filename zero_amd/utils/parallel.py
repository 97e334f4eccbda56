# coding: utf-8
"""Multi-GPU gradient aggregation: one process per GPU, RCCL all-reduce over xGMI.

Counterpart of the reference's utils/parallel.py:79-208.  The reference replicates
"towers" inside one TF graph, shards the variables parameter-server style across the GPUs
(parallel.py:105-110) and averages per-variable gradients with concat + reduce_mean
(parallel.py:184-196).  Here every rank holds a full replica (zero_amd/variables.py) and
the only exchange is a sum all-reduce of the flat fp32 gradient buffer, issued bucket by
bucket while the backward is still running (torch.distributed's ``nccl`` backend *is* RCCL
on ROCm; collectives run on RCCL's own stream and are ordered after the compute stream's
position at call time, so they overlap with the rest of the backward).  The 1/N of
``average_gradients`` is folded into the optimizer pass (utils/cycle.py).

Equality with the reference's tower average: mean over ranks of per-rank gradients
(parallel.py:196) -- identical to the big-batch gradient only when every rank holds the
same number of sentences (loss is a mean of per-sentence means, transformer.py:210-211).
"""

import ctypes
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  Single process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # ZERO_DIST_BACKEND=gloo + ZERO_SINGLE_DEVICE=1: several ranks on ONE GPU (RCCL refuses that), the
            # way the GPU tests and a 1-GPU box exercise the multi-rank step end to end
            backend = os.environ.get("ZERO_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("ZERO_SINGLE_DEVICE", "0") != "0":
            local = 0
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def layer_buckets(store):
    """Contiguous [lo, hi) element ranges of the flat gradient buffer in the order the
    backward finishes them: softmax/target embedding, decoder layers (last first), encoder
    layers (last first), source embedding + shared bias."""
    names = store.names()
    ends = {}
    order = []
    for i, n in enumerate(names):
        lo = store.offsets[n]
        hi = store.offsets[names[i + 1]] if i + 1 < len(names) else store.numel
        parts = n.split("/")
        key = "/".join(parts[:2]) if parts[0] in ("encoder", "decoder") else n
        if key not in ends:
            ends[key] = [lo, hi]
            order.append(key)
        else:
            ends[key][1] = hi
    return {k: tuple(v) for k, v in ends.items()}, order


class RcclComm(object):
    """RCCL through the C-ABI (``zk_comm_*``, zero_amd/csrc/zk_comm.hip): the communicator handle is owned by this
    object; collectives are enqueued on a side HIP stream of its own, ordered after the compute stream's
    position at call time, and hand back a HIP event the consumer stream waits on.

    The rendezvous uses whatever ``torch.distributed`` group is up (gloo is enough): rank 0 draws the 128-byte id,
    ``broadcast_object_list`` ships it.  With a single process the id stays local (used by the GPU test)."""

    def __init__(self, device=None):
        from zero_amd import hip
        self.lib = hip.lib()
        if not self.lib.raw("zk_comm_available")():
            raise hip.ZeroHipError("librccl is not loadable: %s" % self.lib.raw("zk_last_error_string")().decode())
        self.rank, self.world = rank(), world_size()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        uid = ctypes.create_string_buffer(128)
        if self.rank == 0:
            self.lib.call("zk_comm_unique_id", uid)
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(box[0], 128)
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self.lib.call("zk_comm_init", uid, self.world, self.rank, ctypes.byref(self.handle))
        self.stream = torch.cuda.Stream(self.device)
        self.calls = 0

    def _after_compute(self):
        """Order this communicator's stream behind the compute stream's position.  Default: an event.
        ZERO_HIP_COMM_HANDOFF=flag (round 6): a monotonic device word instead -- the compute stream adds one (a one-thread
        launch), a one-thread poll kernel on the communication stream waits for the count; no event is recorded on the
        compute stream (the single-rank loop measured 40-65 us per step for one such event).  Same
        ordering, same collectives: results are bit-identical (tests/test_gpu_dp.py)."""
        cur = torch.cuda.current_stream(self.device)
        if os.environ.get("ZERO_HIP_COMM_HANDOFF", "event").lower() == "flag":
            if getattr(self, "_flag", None) is None:
                self._flag = torch.zeros(2, dtype=torch.int64, device=self.device)       # [count, error word]
                self._handoffs = 0
                torch.cuda.current_stream(self.device).synchronize()
            self._handoffs += 1
            self.lib.call("zk_flag_add", self._flag.data_ptr(), cur.cuda_stream)
            self.lib.call("zk_flag_wait", self._flag.data_ptr(), self._handoffs, self._flag.data_ptr() + 8,
                          self.stream.cuda_stream)
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        self.stream.wait_event(ev)

    def handoff_errors(self):
        """Number of flag hand-offs that gave up waiting (0 unless the device is wedged); forces a sync."""
        f = getattr(self, "_flag", None)
        return 0 if f is None else int(f[1:2].cpu()[0] & 0xffffffff)

    def all_reduce(self, t):
        """In-place sum over the ranks of a contiguous fp32 / bf16 tensor; returns the completion event."""
        self.calls += 1
        assert t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        self._after_compute()
        self.lib.call("zk_comm_allreduce", self.handle, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1,
                      self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done

    def all_gather(self, send, recv):
        code = {torch.float32: 0, torch.bfloat16: 1, torch.int32: 2}[send.dtype]
        assert recv.numel() == send.numel() * self.world and recv.dtype == send.dtype
        self._after_compute()
        self.lib.call("zk_comm_allgather", self.handle, send.data_ptr(), recv.data_ptr(), send.numel(), code,
                      self.stream.cuda_stream)
        done = torch.cuda.Event()
        done.record(self.stream)
        return done

    def close(self):
        if self.handle is not None and self.handle.value:
            self.stream.synchronize()
            self.lib.call("zk_comm_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _EventWork(object):
    """torch.distributed Work look-alike over a HIP event (the RCCL transport)."""

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


_TRANSPORT = {}


def _all_ranks_ok(ok):
    """Collective AND of a local success flag over the torch.distributed group (so that either EVERY rank takes the
    direct transport or none does: a rank falling back on its own would leave the others blocked in a collective)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.cpu()[0]))


def _transport_self_test(t):
    """One small sum all-reduce (fp32), one (bf16) and one all-gather through the direct communicator, checked against
    values every rank can compute locally (rank-dependent integers, exact in fp32 / bf16).  EVERY rank issues all
    three collectives whatever it saw in the earlier ones (a rank that returned early on a mismatch would leave the
    others blocked in the next collective); the verdicts are combined afterwards."""
    r, w = t.rank, t.world
    dev = t.device
    x = torch.arange(1024, dtype=torch.float32, device=dev) % 7 + float(r)
    t.all_reduce(x).synchronize()
    want = (torch.arange(1024, dtype=torch.float32, device=dev) % 7) * w + float(w * (w - 1) // 2)
    ok = bool(torch.equal(x, want))
    xb = torch.full((512,), float(r + 1), dtype=torch.bfloat16, device=dev)
    t.all_reduce(xb).synchronize()
    ok = ok and float(xb.float().max()) == float(w * (w + 1) // 2) and float(xb.float().min()) == float(w * (w + 1) // 2)
    send = torch.full((256,), r, dtype=torch.int32, device=dev)
    recv = torch.empty(256 * w, dtype=torch.int32, device=dev)
    t.all_gather(send, recv).synchronize()
    return ok and bool(torch.equal(recv.view(w, 256)[:, 0].cpu(), torch.arange(w, dtype=torch.int32)))


def _bring_up_direct(log):
    """Collective attempt to create the direct communicator: (RcclComm or None, reason).  Order of the collectives is
    the same on every rank whatever happens locally: availability AND -> id broadcast + ncclCommInitRank -> success AND
    (before any collective of the new communicator) -> self-test (all three collectives on every rank) -> verdict AND."""
    from zero_amd import hip
    avail = bool(hip.lib().raw("zk_comm_available")())
    if not _all_ranks_ok(avail):
        return None, "librccl not loadable on every rank"
    t, why = None, None
    try:
        t = RcclComm()
    except Exception as exc:      # noqa: BLE001 -- never lose the job to the optional transport
        why = repr(exc)
    if not _all_ranks_ok(t is not None):
        why = why or "another rank failed to create its communicator"
    else:
        try:
            good = _transport_self_test(t)
        except Exception as exc:  # noqa: BLE001
            good, why = False, repr(exc)
        if not _all_ranks_ok(good):
            why = why or "self-test mismatch (here or on another rank)"
    if why is not None and t is not None:
        try:
            t.close()
        except Exception:         # noqa: BLE001
            pass
        t = None
    return t, why


def transport():
    """Which library call carries the gradient buckets.  Decided by rule, once per process, COLLECTIVELY:

      ZERO_HIP_COMM=torch (default) ``torch.distributed`` (whose ``nccl`` backend IS RCCL on ROCm; the only choice for
                                    gloo and for several ranks sharing one GPU).  The default until a multi-GPU run
                                    has carried gradients over the direct transport: `bench.py --gpus N` times both
                                    and prints them side by side (``rccl.legs``).
      ZERO_HIP_COMM=auto            the C-ABI communicator (``zk_comm_*`` = RCCL on a side HIP stream of its own) when
                                    librccl is loadable, world > 1 and every rank owns its own GPU (torch.distributed
                                    backend ``nccl``); otherwise ``torch.distributed``
      ZERO_HIP_COMM=rccl            the same attempt, also with a single process

    After the local attempt (dlopen, ncclCommInitRank, a small all-reduce / all-gather self-test) the ranks
    all-reduce(MIN) their success flags over the torch.distributed group: the direct transport is used only if it came
    up on EVERY rank; otherwise every rank destroys its communicator and logs the fallback.  Both transports issue the
    same sum all-reduces / all-gathers of the same buffers."""
    if "t" not in _TRANSPORT:
        mode = os.environ.get("ZERO_HIP_COMM", "torch").lower()
        _TRANSPORT["t"] = _decide_transport(mode)
    return _TRANSPORT["t"] if _TRANSPORT.get("use", True) else None


def _decide_transport(mode):
    import logging
    log = logging.getLogger("zero_amd")
    world = world_size()
    distinct = dist.is_initialized() and dist.get_backend() == "nccl" and os.environ.get("ZERO_SINGLE_DEVICE", "0") == "0"
    want = torch.cuda.is_available() and ((mode == "auto" and world > 1 and distinct) or
                                          (mode == "rccl" and (world == 1 or distinct)))
    if not want:
        return None
    t, why = _bring_up_direct(log)
    if why is not None:
        log.warning("direct RCCL transport (zk_comm) NOT used on rank %d: %s; every rank falls back to "
                    "torch.distributed", rank(), why)
    return t


def select_transport(direct):
    """Measurement aid (bench.py --gpus N): switch between the direct communicator and torch.distributed at run time.
    COLLECTIVE: every rank must call it with the same argument at the same point.  direct=True brings the communicator
    up on first use (collectively, with the self-test); returns whether the direct transport is now in use."""
    if direct:
        if _TRANSPORT.get("t") is None:
            _TRANSPORT["t"] = _decide_transport("auto")
        _TRANSPORT["use"] = True
        return _TRANSPORT["t"] is not None
    _TRANSPORT.setdefault("t", None)
    _TRANSPORT["use"] = False
    return False


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class HipBucketOps(object):
    """Device side of the exchange (casts of a bucket, pack / scatter of table rows): HIP kernels through the C-ABI on
    the current stream.  The host logic of GradientAllReduce only talks to this interface; the CPU tests of that logic
    plug in a torch stand-in (tests/test_parallel_gloo.py), the product never does."""

    def __init__(self):
        from zero_amd import hip
        self.lib = hip.lib()

    @staticmethod
    def _s(t):
        return torch.cuda.current_stream(t.device).cuda_stream

    def cast_to_bf16(self, src, dst):
        self.lib.call("zk_cast_f32_bf16", src.data_ptr(), dst.data_ptr(), src.numel(), self._s(src))

    def cast_to_f32(self, src, dst):
        self.lib.call("zk_cast_bf16_f32", src.data_ptr(), dst.data_ptr(), src.numel(), self._s(src))

    def payload_words(self, R, H, bf16):
        return int(self.lib.query("zk_rows_payload_bytes", R, H, 1 if bf16 else 0)) // 4

    def rows_pack(self, table, uid, n_dev, out, R, H, bf16):
        self.lib.call("zk_rows_pack", table.data_ptr(), uid.data_ptr(), n_dev.data_ptr(), out.data_ptr(), R, H,
                      1 if bf16 else 0, 1, self._s(table))

    def rows_scatter_add(self, table, payload, R, H, bf16, vocab_rows):
        self.lib.call("zk_rows_scatter_add", table.data_ptr(), payload.data_ptr(), R, H, 1 if bf16 else 0, vocab_rows,
                      self._s(table))


class GradientAllReduce(object):
    """Bucketed, overlapped exchange of ``store.grad`` between the data-parallel ranks.

    * dense buckets: sum all-reduce of contiguous ranges of the flat fp32 gradient buffer, handed over as the backward
      finishes them.  ``bucket_dtype`` fp32 by default -- the reference's tower average is fp32
      (utils/parallel.py:184-196); ``ZERO_HIP_BUCKET_DTYPE=bf16`` (opt-in): the range is cast into a bf16 staging
      buffer, all-reduced there (half the bytes on xGMI) and cast back into the fp32 gradient buffer before its Adam
      pass, which accumulates in fp32 as before.
    * row-sparse tables (``set_sparse``): an embedding table that is only ever looked up receives gradient rows for
      the <= T ids of the batch; the ranks all-gather their packed (ids, rows) payloads and every rank adds all N
      payloads into its zeroed table in rank order (utils/parallel.py:142-181: IndexedSlices concatenated across the
      towers, then de-duplicated).  8 ranks x 4096 rows x 512 x 2 B = 33.6 MB received per rank instead of the
      2 x 7/8 x 65.5 MB a ring all-reduce of the dense fp32 table moves.
    """

    def __init__(self, store, bucket_elems=8 * 1024 * 1024, ops=None, bucket_dtype=None):
        self.store = store
        self.bucket_elems = bucket_elems
        self.ranges, _ = layer_buckets(store)
        self.pending = []
        self._open = None   # (lo, hi) of the bucket being filled (adjacent ranges only)
        on_gpu = store.grad.is_cuda
        self.ops = ops if ops is not None else (HipBucketOps() if on_gpu else None)
        if bucket_dtype is None:
            # fp32 = what the reference averages (utils/parallel.py:184-196) and the default; bf16 halves the bytes
            # on xGMI and is opt-in until a multi-GPU run shows that the fp32 exchange is exposed (bench.py --gpus N
            # times both: rccl.legs)
            bucket_dtype = os.environ.get("ZERO_HIP_BUCKET_DTYPE", "fp32")
        if isinstance(bucket_dtype, str):
            bucket_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[bucket_dtype.lower()]
        if bucket_dtype == torch.bfloat16 and self.ops is None:
            raise ValueError("bf16 gradient buckets need the device-side cast ops")
        self.bucket_dtype = bucket_dtype
        self._stage = None          # bf16 staging buffer, same length as the gradient buffer (allocated on first use)
        self._sparse = {}           # key -> dict(table, uid, n, R, H, V) for the CURRENT batch
        self._payload = {}          # (key, "send" | "recv") -> int32 buffer
        self.sparse_enabled = os.environ.get("ZERO_HIP_SPARSE_EMBED", "1") != "0" and self.ops is not None
        self.disabled = False       # measurement aid (bench.py): no exchange at all, buckets are only bookkept
        self.bytes_step = 0         # bytes this rank hands to the collectives in the current step
        self.bytes_last_step = 0

    # -- reporting ---------------------------------------------------------------------------------------------
    def bucket_dtype_name(self):
        return "bf16" if self.bucket_dtype == torch.bfloat16 else "fp32"

    def sparse_keys(self):
        return sorted(self._sparse)

    # -- row-sparse tables ---------------------------------------------------------------------------------------
    def set_sparse(self, key, table, uid, n_dev, capacity, rows=None):
        """Declare ``key`` (a variable of the flat buffer that is a pure lookup table) row-sparse for the current batch:
        ``uid`` the sorted distinct ids the batch touched, ``n_dev`` their count (device int32), ``rows`` the number of
        token rows of the batch (an upper bound of the count), ``capacity`` the slot count of a payload -- the SAME on
        every rank (derived from the batching limits, Trainer._sparse_capacity)."""
        if not self.sparse_enabled or world_size() == 1:
            return
        V, H = table.shape
        R = (int(capacity) + 3) // 4 * 4
        if R * world_size() >= 2 * V:
            # a ring all-gather moves (N-1) x R rows per rank, a ring all-reduce of the dense table 2 (N-1)/N x V rows:
            # the payloads only pay while R < 2 V / N (same decision on every rank: it depends on the limits alone)
            return
        if rows is not None and rows > R and not getattr(self, "_warned_rows", False):
            # The payload holds one slot per DISTINCT id, the batch has `rows` token rows: more rows than slots is fine
            # as long as the ids repeat enough.  Whether they do is only known on the device (zk_batch_prep counts
            # them), and a rank must not raise on its own -- the others would wait in the all-gather for ever.  So the
            # decision is the device's: zk_rows_pack poisons its payload with NaN when the count exceeds the capacity,
            # every rank adds the same NaN into its table, and every rank stops at its next NaN check (main.py:316-319).
            import logging
            logging.getLogger("zero_amd").warning(
                "sparse exchange of %s: the batch has %d token rows, payload capacity is %d distinct ids; if the batch "
                "touches more ids than that the gradient is poisoned with NaN on every rank (raise token_size or set "
                "ZERO_HIP_SPARSE_EMBED=0)", key, rows, R)
            self._warned_rows = True
        self._sparse[key] = {"table": table, "uid": uid, "n": n_dev, "R": R, "H": H, "V": V}

    def clear_sparse(self):
        self._sparse = {}

    def configure(self, bucket_dtype=None, sparse=None):
        """Measurement aid (bench.py --gpus N: one process group times several exchange modes).  COLLECTIVE in effect:
        every rank must make the same change between two steps (nothing may be pending)."""
        assert not self.pending and self._open is None, "exchange in flight"
        if bucket_dtype is not None:
            bucket_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[bucket_dtype] \
                if isinstance(bucket_dtype, str) else bucket_dtype
            if bucket_dtype == torch.bfloat16 and self.ops is None:
                raise ValueError("bf16 gradient buckets need the device-side cast ops")
            self.bucket_dtype = bucket_dtype
        if sparse is not None:
            self.sparse_enabled = bool(sparse) and self.ops is not None
            if not self.sparse_enabled:
                self._sparse = {}

    def _buf(self, key, kind, words, device):
        b = self._payload.get((key, kind))
        if b is None or b.numel() != words:
            b = torch.empty(words, dtype=torch.int32, device=device)
            self._payload[(key, kind)] = b
        return b

    def _exchange_rows(self, key):
        sp = self._sparse[key]
        w = world_size()
        bf16 = self.bucket_dtype == torch.bfloat16
        table, R, H = sp["table"], sp["R"], sp["H"]
        words = self.ops.payload_words(R, H, bf16)
        send = self._buf(key, "send", words, table.device)
        recv = self._buf(key, "recv", words * w, table.device)
        self.ops.rows_pack(table, sp["uid"], sp["n"], send, R, H, bf16)
        tr = transport()
        if tr is not None:
            work = _EventWork(tr.all_gather(send, recv))
        elif dist.get_backend() == "nccl":
            work = dist.all_gather_into_tensor(recv, send, async_op=True)
        else:
            # gloo (the CPU tests; the GPU tests with several ranks on one device): blocking, staged through the host
            host = send.cpu()
            outs = [torch.empty_like(host) for _ in range(w)]
            dist.all_gather(outs, host)
            for r in range(w):
                recv[r * words:(r + 1) * words].copy_(outs[r])
            work = None
        self.bytes_step += words * 4

        def post():
            for r in range(w):      # fixed order: bit-identical tables on every rank
                self.ops.rows_scatter_add(table, recv[r * words:(r + 1) * words], R, H, bf16, sp["V"])
        lo, hi = self.ranges[key]
        self.pending.append((lo, hi, work, post))

    # -- dense buckets -------------------------------------------------------------------------------------------
    def _all_reduce(self, t):
        tr = transport()
        if tr is not None:
            return _EventWork(tr.all_reduce(t))
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def _exchange_range(self, lo, hi):
        g = self.store.grad[lo:hi]
        if self.bucket_dtype == torch.float32:
            self.bytes_step += (hi - lo) * 4
            return self._all_reduce(g), None
        if self._stage is None:
            self._stage = torch.empty(self.store.numel, dtype=torch.bfloat16, device=g.device)
        st = self._stage[lo:hi]
        self.ops.cast_to_bf16(g, st)
        self.bytes_step += (hi - lo) * 2
        return self._all_reduce(st), (lambda: self.ops.cast_to_f32(st, g))

    def ready(self, key):
        """Called by the backward when every gradient under ``key`` is final."""
        if world_size() == 1:
            return
        lo, hi = self.ranges[key]
        if self.disabled:
            self.pending.append((lo, hi, None, None))
            return
        if key in self._sparse:
            self.flush()
            self._exchange_rows(key)
            return
        if self._open is not None and (self._open[0] == hi or self._open[1] == lo):
            self._open = (min(lo, self._open[0]), max(hi, self._open[1]))
        else:
            self.flush()
            self._open = (lo, hi)
        if self._open[1] - self._open[0] >= self.bucket_elems:
            self.flush()

    def flush(self):
        if self._open is None or world_size() == 1:
            self._open = None
            return
        lo, hi = self._open
        self._open = None
        work, post = self._exchange_range(lo, hi)
        self.pending.append((lo, hi, work, post))

    def wait(self):
        """Flush the tail and make the current stream wait for every bucket."""
        for _ in self.drain():
            pass

    def drain(self):
        """Flush the tail, then yield (lo, hi) of every bucket in launch order as soon as the current stream
        has been made to wait for ITS exchange (and the cast back / row scatter of that bucket is enqueued): work
        queued per bucket (the Adam update of that range) overlaps the collectives still in flight."""
        self.flush()
        pending, self.pending = self.pending, []
        self.bytes_last_step, self.bytes_step = self.bytes_step, 0
        for lo, hi, w, post in pending:
            if w is not None:
                w.wait()
            if post is not None:
                post()
            yield lo, hi

    def all_reduce_everything(self):
        """Unbucketed path (used after gradient accumulation: one exchange per update,
        cycle.py:86-88 semantics).  Dense and fp32: the accumulated table gradients are no longer row-sparse."""
        if world_size() > 1:
            self._all_reduce(self.store.grad).wait()


def average_scalar(t):
    """Tower-mean of a scalar (main.py:42 loss mean) -- logging only."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t
