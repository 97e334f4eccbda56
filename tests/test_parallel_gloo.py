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
