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

    def _after_compute(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(ev)

    def all_reduce(self, t):
        """In-place sum over the ranks of a contiguous fp32 / bf16 tensor; returns the completion event."""
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


def transport():
    """``ZERO_HIP_COMM=rccl``: collectives go through the C-ABI communicator (``zk_comm_*``) on its own side
    stream; default ``torch``: ``torch.distributed`` (whose ``nccl`` backend IS RCCL on ROCm, and which is the only
    choice for the gloo tests).  Both are sum all-reduces of the same buckets; results are identical.  The direct
    transport has only ever been exercised with ONE rank on the 1-GPU test box (tests/test_gpu_dp.py), so it is
    opt-in until an 8-GPU run has seen it; any failure to set it up falls back to torch.distributed."""
    if "t" not in _TRANSPORT:
        t = None
        if os.environ.get("ZERO_HIP_COMM", "torch").lower() == "rccl" and torch.cuda.is_available() \
                and (not dist.is_initialized() or dist.get_backend() != "gloo" or os.environ.get("ZERO_SINGLE_DEVICE", "0") == "0"):
            try:
                t = RcclComm()
            except Exception as exc:      # noqa: BLE001 -- never lose the job to the optional transport
                import logging
                logging.getLogger("zero_amd").warning("direct RCCL transport unavailable (%s); torch.distributed", exc)
        _TRANSPORT["t"] = t
    return _TRANSPORT["t"]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GradientAllReduce(object):
    """Bucketed, overlapped sum all-reduce of ``store.grad``."""

    def __init__(self, store, bucket_elems=8 * 1024 * 1024):
        self.store = store
        self.bucket_elems = bucket_elems
        self.ranges, _ = layer_buckets(store)
        self.pending = []
        self._open = None   # (lo, hi) of the bucket being filled (adjacent ranges only)

    def _all_reduce(self, t):
        tr = transport()
        if tr is not None:
            return _EventWork(tr.all_reduce(t))
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    def ready(self, key):
        """Called by the backward when every gradient under ``key`` is final."""
        if world_size() == 1:
            return
        lo, hi = self.ranges[key]
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
        self.pending.append((lo, hi, self._all_reduce(self.store.grad[lo:hi])))

    def wait(self):
        """Flush the tail and make the current stream wait for every bucket."""
        for _ in self.drain():
            pass

    def drain(self):
        """Flush the tail, then yield (lo, hi) of every bucket in launch order as soon as the current stream
        has been made to wait for ITS all-reduce: work queued per bucket (the Adam update of that range)
        overlaps the collectives still in flight."""
        self.flush()
        pending, self.pending = self.pending, []
        for lo, hi, w in pending:
            w.wait()
            yield lo, hi

    def all_reduce_everything(self):
        """Unbucketed path (used after gradient accumulation: one exchange per update,
        cycle.py:86-88 semantics)."""
        if world_size() > 1:
            self._all_reduce(self.store.grad).wait()


def average_scalar(t):
    """Tower-mean of a scalar (main.py:42 loss mean) -- logging only."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t /= world_size()
    return t
