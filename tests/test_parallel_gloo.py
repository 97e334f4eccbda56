"""N>1 path on CPU: two gloo ranks run the bucketed gradient all-reduce of
zero_amd/utils/parallel.py over a CPU-resident variable store and must end up with the
tower SUM (the 1/N is folded into the optimizer scale) -- utils/parallel.py:134-208."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import make_hp
from zero_amd.utils import parallel
from zero_amd.variables import VariableStore


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _backward_order(store, hp):
    order = ["decoder/layer_%d" % l for l in reversed(range(hp.num_decoder_layer))]
    order += ["tgt_embedding"]
    order += ["encoder/layer_%d" % l for l in reversed(range(hp.num_encoder_layer))]
    order += ["bias", "src_embedding"]
    return order


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world) and parallel.world_size() == world
    hp = make_hp("transformer", H=16, F=32, heads=2, layers=3, Vs=13, Vt=11)
    store = VariableStore(hp, "transformer", "cpu")
    gen = torch.Generator().manual_seed(100 + rank)
    store.grad.copy_(torch.randn(store.numel, generator=gen))
    mine = store.grad.clone()
    red = parallel.GradientAllReduce(store, bucket_elems=4000)
    for key in _backward_order(store, hp):
        red.ready(key)
    red.wait()
    others = [torch.randn(store.numel, generator=torch.Generator().manual_seed(100 + k)) for k in range(world)]
    expect = sum(others)
    ok = bool(torch.allclose(store.grad, expect, atol=1e-6)) and not torch.equal(mine, store.grad)
    loss = parallel.average_scalar(torch.tensor([float(rank)]))
    ok = ok and abs(float(loss) - (world - 1) / 2.0) < 1e-6
    # unbucketed path used with gradient accumulation
    store.grad.copy_(mine)
    red.all_reduce_everything()
    ok = ok and bool(torch.allclose(store.grad, expect, atol=1e-6))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_is_the_tower_sum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_buckets_tile_the_flat_buffer():
    hp = make_hp("transformer_aan", H=16, F=32, heads=2, layers=2, Vs=13, Vt=11)
    store = VariableStore(hp, "transformer_aan", "cpu")
    ranges, order = parallel.layer_buckets(store)
    spans = sorted(ranges.values())
    assert spans[0][0] == 0 and spans[-1][1] == store.numel
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c
    assert "decoder/layer_1" in ranges and "encoder/layer_0" in ranges and "bias" in ranges


def test_single_process_is_a_noop():
    hp = make_hp("transformer", H=16, F=32, heads=2, layers=1, Vs=13, Vt=11)
    store = VariableStore(hp, "transformer", "cpu")
    store.grad.fill_(1.0)
    red = parallel.GradientAllReduce(store)
    red.ready("bias"); red.wait()
    assert float(store.grad.sum()) == store.numel and parallel.world_size() == 1


# ---------------------------------------------------------------------------------------------
# bf16 buckets and the row-sparse table exchange: the HOST logic of GradientAllReduce on CPU tensors.  The
# device side (casts, row pack / scatter) is the HIP kernel set behind parallel.HipBucketOps; this torch
# stand-in with the same interface exists only here (the GPU twin of this test is tests/test_gpu_dp.py).
# ---------------------------------------------------------------------------------------------
class TorchBucketOps(object):
    def cast_to_bf16(self, src, dst):
        dst.copy_(src.to(torch.bfloat16))

    def cast_to_f32(self, src, dst):
        dst.copy_(src.float())

    def payload_words(self, R, H, bf16):
        return R + R * H // (2 if bf16 else 1)

    def rows_pack(self, table, uid, n_dev, out, R, H, bf16):
        n = min(int(n_dev[0]), R)
        ids = torch.full((R,), -1, dtype=torch.int32)
        ids[:n] = uid[:n]
        rows = torch.zeros(R, H)
        rows[:n] = table[uid[:n].long()]
        table[uid[:n].long()] = 0.0
        out[:R] = ids
        out[R:] = (rows.to(torch.bfloat16) if bf16 else rows).reshape(-1).view(torch.int32)

    def rows_scatter_add(self, table, payload, R, H, bf16, vocab_rows):
        ids = payload[:R]
        rows = payload[R:].view(torch.bfloat16 if bf16 else torch.float32).view(R, H).float()
        ok = (ids >= 0) & (ids < vocab_rows)
        table[ids[ok].long()] += rows[ok]


def _sparse_worker(rank, world, port, q, dtype):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    hp = make_hp("transformer", H=16, F=32, heads=2, layers=2, Vs=41, Vt=11)
    store = VariableStore(hp, "transformer", "cpu")

    def local_grad(r):
        gen = torch.Generator().manual_seed(200 + r)
        g = torch.randn(store.numel, generator=gen)
        # the source table only has rows for the ids of rank r's "batch"
        lo, hi = parallel.layer_buckets(store)[0]["src_embedding"]
        Vp = (hi - lo) // 16
        ids = torch.unique(torch.randint(1, 41, (12,), generator=gen)).to(torch.int32)
        t = torch.zeros(Vp, 16)
        t[ids.long()] = torch.randn(len(ids), 16, generator=gen)
        g[lo:hi] = t.reshape(-1)
        return g, ids

    g, ids = local_grad(rank)
    store.grad.copy_(g)
    red = parallel.GradientAllReduce(store, bucket_elems=3000, ops=TorchBucketOps(), bucket_dtype=dtype)
    red.sparse_enabled = True
    uid = torch.zeros(16, dtype=torch.int32)
    uid[:len(ids)] = ids
    red.set_sparse("src_embedding", store.g("src_embedding"), uid, torch.tensor([len(ids)], dtype=torch.int32), 16,
                   rows=12)
    for key in _backward_order(store, hp):
        red.ready(key)
    covered = sum(hi - lo for lo, hi in red.drain())
    alls = [local_grad(k)[0] for k in range(world)]
    if dtype == "fp32":
        expect = sum(alls)
        ok = bool(torch.allclose(store.grad, expect, atol=1e-6))
    else:
        # dense ranges: sum of the bf16-rounded gradients, itself rounded to bf16 (the collective's dtype); table rows:
        # fp32 sum of the bf16-rounded rows in rank order
        r16 = [a.to(torch.bfloat16) for a in alls]
        dense = (r16[0] + r16[1]).float()
        lo, hi = red.ranges["src_embedding"]
        table = sum(a[lo:hi].to(torch.bfloat16).float() for a in alls)
        expect = dense.clone()
        expect[lo:hi] = table
        ok = bool(torch.equal(store.grad, expect))
    ok = ok and covered == store.numel and red.sparse_keys() == ["src_embedding"]
    # payload bytes: ids + rows of the capacity, not the dense table
    lo, hi = red.ranges["src_embedding"]
    es = 4 if dtype == "fp32" else 2
    ok = ok and red.bytes_last_step == (store.numel - (hi - lo)) * es + 16 * 4 + 16 * 16 * es
    q.put((rank, ok, store.grad.clone().numpy().tobytes()))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_rank_sparse_rows_and_bucket_dtype(dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, 2, port, q, dtype)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted((r, ok) for r, ok, _ in res) == [(0, True), (1, True)]
    assert res[0][2] == res[1][2]           # replicas hold bit-identical reduced gradients


def test_transport_rule_without_gpu_is_torch_distributed():
    parallel._TRANSPORT.clear()
    assert parallel.transport() is None     # no GPU / no process group: torch.distributed (or nothing) carries the buckets
    parallel._TRANSPORT.clear()


# ---------------------------------------------------------------------------------------------
# round 4: run-time reconfiguration (bench.py --gpus N times one leg per exchange mode in ONE process group), the
# transport selector, and the payload-capacity rule that must not raise on one rank
# ---------------------------------------------------------------------------------------------
def test_reducer_defaults_and_runtime_configuration():
    hp = make_hp("transformer", H=16, F=32, heads=2, layers=1, Vs=13, Vt=11)
    store = VariableStore(hp, "transformer", "cpu")
    red = parallel.GradientAllReduce(store, ops=TorchBucketOps())
    assert red.bucket_dtype_name() == "fp32"                      # the reference averages fp32 (utils/parallel.py:184-196)
    red.configure(bucket_dtype="bf16", sparse=True)
    assert red.bucket_dtype_name() == "bf16" and red.sparse_enabled
    red.configure(bucket_dtype="fp32", sparse=False)
    assert red.bucket_dtype_name() == "fp32" and not red.sparse_enabled and red.sparse_keys() == []
    bare = parallel.GradientAllReduce(store)                      # no device-side cast ops on CPU
    with pytest.raises(ValueError):
        bare.configure(bucket_dtype="bf16")


def test_select_transport_is_collective_safe_without_gpus():
    parallel._TRANSPORT.clear()
    assert parallel.select_transport(True) is False               # no GPU / one process: the direct communicator is never tried
    assert parallel.transport() is None
    assert parallel.select_transport(False) is False and parallel.transport() is None
    parallel._TRANSPORT.clear()
    assert os.environ.get("ZERO_HIP_COMM", "torch") == "torch"    # default: torch.distributed (whose nccl backend is RCCL)


def _overflow_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_distributed("gloo")
    hp = make_hp("transformer", H=16, F=32, heads=2, layers=1, Vs=41, Vt=11)
    store = VariableStore(hp, "transformer", "cpu")
    red = parallel.GradientAllReduce(store, ops=TorchBucketOps())
    red.sparse_enabled = True
    uid = torch.zeros(32, dtype=torch.int32)
    # rank 1's batch has MORE token rows than the payload has slots: no exception (the others would wait in the all-gather
    # for ever); the decision is the device's (zk_rows_pack poisons the payload when the DISTINCT ids exceed the capacity)
    try:
        red.set_sparse("src_embedding", store.g("src_embedding"), uid, torch.tensor([3], dtype=torch.int32), 8,
                       rows=8 if rank == 0 else 30)
        ok = red.sparse_keys() == ["src_embedding"]
    except Exception:       # noqa: BLE001
        ok = False
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


def test_payload_capacity_never_raises_on_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
