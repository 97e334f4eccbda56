# coding: utf-8
"""Batch feed either side of the step: the reference's ``EnQueuer``
(utils/queuer.py:36-113) and the host->HBM hand-off the TF1 ``feed_dict`` did implicitly.

Same constructor and iteration contract as the reference:
``EnQueuer(reader, preprocessor, worker_processes_num, input_queue_size, output_queue_size)``;
iterating yields ``preprocessor(chunk)`` for every chunk of ``reader``;
``worker_processes_num == 0`` runs inline; ``1`` runs reader+preprocessor on one background
worker (order preserved); ``n > 1`` uses one reader worker and ``n - 1`` preprocessing workers
sharing an input queue (order not guaranteed, exactly like utils/queuer.py:70-89); a negative
count raises ``ValueError``.

MI355X-side design choice: the workers are *threads* of the rank's process, not forked
processes.  One process per GPU already owns a HIP context, and forking a process with a live
HIP/RCCL context is undefined; the batcher's work is numpy slicing (GIL released in the copies),
and the queue only has to stay ahead of a ~6 ms step.  :class:`DeviceFeeder` then stages each
batch through two pinned host buffers and a copy stream so the H2D of batch i+1 overlaps step i.
"""

import queue as _queue
import threading

TERMINATION_TOKEN = "<DONE>"


class _Failure(object):
    def __init__(self, exc):
        self.exc = exc


class _Worker(threading.Thread):
    def __init__(self, chunks, out):
        super(_Worker, self).__init__()
        self.daemon = True
        self._chunks, self._out = chunks, out

    def run(self):
        try:
            for chunk in self._chunks:
                self._out.put(chunk)
        except BaseException as exc:           # surfaces in the consumer instead of a silent hang
            self._out.put(_Failure(exc))
        self._out.put(TERMINATION_TOKEN)


def _is_term(x):
    return isinstance(x, str) and x == TERMINATION_TOKEN


def _drain(q):
    """Chunks of ``q`` until the termination token, which is put back for the sibling workers."""
    while True:
        chunk = q.get()
        if _is_term(chunk):
            q.put(chunk)
            return
        yield chunk


def _apply(chunks, fn):
    for chunk in chunks:
        if isinstance(chunk, _Failure):
            raise chunk.exc
        yield fn(chunk)


class EnQueuer(object):
    def __init__(self, reader, preprocessor, worker_processes_num=1, input_queue_size=5, output_queue_size=5):
        if worker_processes_num < 0:
            raise ValueError("worker_processes_num must be a non-negative integer.")
        self.worker_processes_number = worker_processes_num
        self.preprocessor = preprocessor
        self.input_queue_size = input_queue_size
        self.output_queue_size = output_queue_size
        self.reader = reader

    def __iter__(self):
        if self.worker_processes_number == 0:
            return _apply(self.reader, self.preprocessor)
        return self._threaded()

    def _threaded(self):
        out = _queue.Queue(self.output_queue_size)
        n = self.worker_processes_number
        if n > 1:
            feed = _queue.Queue(self.input_queue_size)
            workers = [_Worker(self.reader, feed)]
            workers += [_Worker(_apply(_drain(feed), self.preprocessor), out) for _ in range(n - 1)]
            expected = n - 1
        else:
            workers = [_Worker(_apply(self.reader, self.preprocessor), out)]
            expected = 1
        for w in workers:
            w.start()
        seen = 0
        while seen < expected:
            chunk = out.get()
            if _is_term(chunk):
                seen += 1
                continue
            if isinstance(chunk, _Failure):
                raise chunk.exc
            yield chunk
        for w in workers:
            w.join()


class PinnedRing(object):
    """Pinned staging slots for the small per-step uploads of the training loop (token ids, optimiser scalars): ``put``
    copies a host array into the next slot and enqueues an ASYNCHRONOUS copy to its device tensor on the current
    stream.  A slot is reused only after the copy that last read it has completed (one event per slot), so the host may
    run up to ``slots`` uploads ahead of the device and blocks -- bounded run-ahead -- beyond that.  (Pageable
    ``tensor.copy_`` uploads, which rounds 1-3 used, stall the host until the device has caught up: every step.)"""

    def __init__(self, device, slots=8):
        import torch
        self._torch = torch
        self.device = torch.device(device)
        self._n = slots
        self._events = [None] * slots
        self._bufs = {}            # dtype -> pinned [slots, capacity]: every slot is allocated at once (an allocation
        self._i = 0                # later on could meet another thread's stream capture, zero_amd/evalu.py lanes)

    def _slot_view(self, i, dtype, n):
        torch = self._torch
        buf = self._bufs.get(dtype)
        if buf is None or buf.shape[1] < n:
            for ev in self._events:                   # the old buffers may still be read by copies in flight
                if ev is not None:
                    ev.synchronize()
            cap = max(4096, 1 << (int(n) - 1).bit_length())
            buf = torch.empty((self._n, cap), dtype=dtype).pin_memory()
            self._bufs[dtype] = buf
        return buf[i, :n]

    def put(self, dst, arr):
        torch = self._torch
        src = torch.as_tensor(arr).reshape(-1)
        flat = dst.reshape(-1)
        n = src.numel()
        assert flat.numel() == n, (flat.numel(), n)
        if n == 0:
            return
        if self.device.type != "cuda":
            flat.copy_(src)
            return
        i = self._i
        self._i = (i + 1) % self._n
        if self._events[i] is not None:
            self._events[i].synchronize()             # the copy that last read this slot is done
        view = self._slot_view(i, dst.dtype, n)
        view.copy_(src)                               # host-side conversion (int64 ids -> int32) into pinned memory
        flat.copy_(view, non_blocking=True)
        ev = self._events[i]
        if ev is None:
            ev = self._events[i] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))


class DeviceFeeder(object):
    """Double-buffered pinned H2D feed of ``{"src","tgt",...}`` batches (the role of TF1's
    feed_dict copy, main.py:287-294).  Yields ``(batch, device_batch)`` where ``device_batch``
    holds int32 ``source`` / ``target`` tensors resident in HBM; the copy of the next batch is in
    flight on a dedicated stream while the caller runs the current step."""

    def __init__(self, batches, device):
        import torch
        self._torch = torch
        self.device = torch.device(device)
        self._it = iter(batches)
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self._pinned = [{}, {}]
        self._slot = 0

    def _stage(self, batch):
        torch = self._torch
        slot = self._pinned[self._slot]
        self._slot ^= 1
        if slot.get("event") is not None:
            slot["event"].synchronize()        # the copy that last read these pinned buffers is done
        out = {}
        for key, name in (("src", "source"), ("tgt", "target")):
            if key not in batch:
                continue
            arr = batch[key]
            if self._stream is None:
                out[name] = torch.as_tensor(arr, dtype=torch.int32)
                continue
            n = arr.size
            buf = slot.get(key)
            if buf is None or buf.numel() < n:
                buf = torch.empty(max(n, 1024), dtype=torch.int32).pin_memory()
                slot[key] = buf
            view = buf[:n].view(arr.shape)
            view.copy_(torch.as_tensor(arr, dtype=torch.int32))
            with torch.cuda.stream(self._stream):
                out[name] = view.to(self.device, non_blocking=True)
        ev = None
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            slot["event"] = ev
        return batch, out, ev

    def __iter__(self):
        nxt = None
        for batch in self._it:
            cur, nxt = nxt, self._stage(batch)
            if cur is not None:
                yield self._ready(cur)
        if nxt is not None:
            yield self._ready(nxt)

    def _ready(self, staged):
        batch, out, ev = staged
        if ev is not None:
            cur = self._torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in out.values():
                t.record_stream(cur)
        return batch, out
