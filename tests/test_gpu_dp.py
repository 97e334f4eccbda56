"""Data-parallel step on the GPU box's single MI355X: two processes (gloo backend, both on cuda:0 --
RCCL refuses two ranks on one device) run the REAL multi-rank step of zero_amd/main.py: hipGraph
segments cut at the gradient-bucket hand-offs, bucketed all-reduce overlapping the backward, 1/N
folded into Adam (utils/parallel.py:134-208, cycle.py:86-101).  Checks: every rank ends with the
same weights; the averaged gradient equals the gradient of the concatenated batch computed by
one rank; the segmented replay equals the eager multi-rank step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, use_graph, overlap="1", exchange=None):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), ZERO_HIP_GROUP_LAYERS="1", ZERO_HIP_OVERLAP_UPDATE=overlap)
    if exchange is not None:         # (bucket dtype, row-sparse tables on / off); default = the product's defaults
        # payload capacity 40 rows: the toy batch has 36 source tokens and a 120-row table (the rule from the
        # batching limits, 3000 rows, would exceed the table and select the dense exchange)
        os.environ.update(ZERO_HIP_BUCKET_DTYPE=exchange[0], ZERO_HIP_SPARSE_EMBED=exchange[1], ZERO_HIP_SPARSE_ROWS="40")
    from tests.common import make_hp, make_batch
    from zero_amd.utils import parallel
    from zero_amd.main import Trainer
    parallel.init_distributed("gloo")
    torch.cuda.set_device(0)
    hp = make_hp("transformer", lrate=0.5, warmup_steps=10)
    hp.random_seed = 7
    rng = np.random.default_rng(50)
    src, tgt = make_batch(rng, 8, 9, 10, hp.src_vocab.size(), hp.tgt_vocab.size(), full_first=False)
    src[:, -1], tgt[:, -1] = 2, 2                 # full-length rows: no column trimming differences between halves
    src[src == 0] = 5; tgt[tgt == 0] = 5
    half = slice(rank * 4, rank * 4 + 4)
    tr = Trainer(hp)
    tr.prepare_static({"source": src[half], "target": tgt[half]})
    losses = []
    grad1 = None
    for _ in range(3):
        losses.append(float(tr.step_static(use_graph=use_graph).cpu()[0]))
        if grad1 is None:
            torch.cuda.synchronize()
            grad1 = tr.store.grad.cpu().numpy().copy()      # the exchanged gradient of step 1 (identical weights in every mode)
    torch.cuda.synchronize()
    g_, p_, bad_ = tr.train_op.stats()
    tag = "" if overlap == "1" else "_plain"
    if exchange is not None:
        tag += "_%s_%s" % exchange
    np.savez(os.path.join(out_dir, "r%d_%d%s.npz" % (rank, int(use_graph), tag)),
             loss=np.array(losses), gnorm=np.array([g_, p_]),
             exchange=np.array([tr.reducer.bucket_dtype_name(), ",".join(tr.reducer.sparse_keys()),
                                str(tr.reducer.bytes_last_step), str(tr.store.numel)]),
             grad=tr.store.grad.cpu().numpy(), grad1=grad1, master=tr.store.master.cpu().numpy(),
             kinds=np.array([k for k, _ in next((v for k, v in tr._graphs.items() if k[0] == "seg"), [])] or ["none"]))
    torch.distributed.destroy_process_group()


def _run(world, out_dir, use_graph, overlap="1", exchange=None):
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, out_dir, use_graph, overlap, exchange))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, "rank failed (exit code %s)" % p.exitcode


def test_two_rank_step_on_one_gpu(tmp_path):
    out = str(tmp_path)
    _run(2, out, True)
    _run(2, out, False)
    _run(2, out, True, overlap="0")
    g = [np.load(os.path.join(out, "r%d_1.npz" % r)) for r in range(2)]
    e = [np.load(os.path.join(out, "r%d_0.npz" % r)) for r in range(2)]
    plain = [np.load(os.path.join(out, "r%d_1_plain.npz" % r)) for r in range(2)]
    # per-bucket updates behind their own all-reduce == wait-for-all, norm, one Adam pass
    assert np.array_equal(g[0]["master"], plain[0]["master"]) and np.array_equal(g[0]["loss"], plain[0]["loss"])
    assert abs(g[0]["gnorm"][0] - plain[0]["gnorm"][0]) <= 1e-6 * plain[0]["gnorm"][0]
    assert "ready" in list(g[0]["kinds"]) and list(g[0]["kinds"])[-1] == "update"
    # replicas stay identical, segmented replay == eager multi-rank step
    assert np.array_equal(g[0]["master"], g[1]["master"]) and np.array_equal(e[0]["master"], e[1]["master"])
    assert np.array_equal(g[0]["master"], e[0]["master"]) and np.array_equal(g[0]["loss"], e[0]["loss"])
    assert np.array_equal(g[0]["grad"], g[1]["grad"])           # the all-reduced (summed) gradient
    assert g[0]["loss"][-1] < g[0]["loss"][0]
    # against ONE rank on the concatenated batch: grad_sum / 2 == gradient of the 8-sentence batch
    from tests.common import make_hp, make_batch
    from zero_amd.main import Trainer
    from zero_amd.models._factory import reset_cores
    from zero_amd.variables import reset_stores
    reset_cores(); reset_stores()
    hp = make_hp("transformer", lrate=0.5, warmup_steps=10)
    hp.random_seed = 7
    rng = np.random.default_rng(50)
    src, tgt = make_batch(rng, 8, 9, 10, hp.src_vocab.size(), hp.tgt_vocab.size(), full_first=False)
    src[:, -1], tgt[:, -1] = 2, 2
    src[src == 0] = 5; tgt[tgt == 0] = 5
    tr = Trainer(hp)
    tr.prepare_static({"source": src, "target": tgt})
    P0 = {k: v.copy() for k, v in tr.store.export("master").items()}       # the weights every run above started from
    l0 = float(tr.step_static(use_graph=False).cpu()[0])
    torch.cuda.synchronize()
    # step 1 of the two-rank job started from the same weights: its per-rank losses average to l0
    both = 0.5 * (g[0]["loss"][0] + g[1]["loss"][0])
    assert abs(both - l0) / abs(l0) < 2e-3, (both, l0)
    # ... and its EXCHANGED gradient (the sum over the ranks; 1 / N is folded into the update) is twice the gradient of
    # the concatenated batch (utils/parallel.py:184-196 averages the towers; both halves hold four sentences), variable
    # by variable: same kernels on both sides, so what differs is the summation order over 4 + 4 against 8 sentences
    g1 = tr.store.grad.cpu().numpy()
    ex = 0.5 * g[0]["grad1"]
    assert ex.shape == g1.shape
    num, den = float(np.linalg.norm(ex - g1)), float(np.linalg.norm(g1))
    assert num / den < 1e-2, (num, den)
    exd, g1d = tr.store.export(torch.from_numpy(ex.copy())), tr.store.export(torch.from_numpy(g1.copy()))
    worst = 0.0
    for name in exd:
        nb = float(np.linalg.norm(g1d[name]))
        if nb > 1e-3 * den:
            worst = max(worst, float(np.linalg.norm(exd[name] - g1d[name])) / nb)
    assert worst < 3e-2, worst
    # ... and the oracle's tower mean (oracle/ref_torch.py: the fp32 restatement of average_gradients over two towers of
    # four sentences) agrees with the exchanged gradient as well as it agrees with any single-rank bf16 gradient
    from oracle import ref_torch as rt
    Pt = rt.to_torch(P0, torch.float32, requires_grad=True)
    tot = 0.0
    for half in (slice(0, 4), slice(4, 8)):
        r = rt.train_fn({"source": torch.tensor(src[half]), "target": torch.tensor(tgt[half])}, hp, Pt, "transformer",
                        training=False)
        (0.5 * r["loss"]).backward()
        tot += 0.5 * float(r["loss"].detach())
    assert abs(tot - both) / abs(tot) < 2e-3
    og = {k: v.grad.numpy() for k, v in Pt.items() if v.grad is not None}
    num = sum(float(((exd[k].astype(np.float64) - og[k]) ** 2).sum()) for k in og)
    den2 = sum(float((og[k].astype(np.float64) ** 2).sum()) for k in og)
    assert (num / den2) ** 0.5 < 6e-2, (num / den2) ** 0.5


def test_exchange_modes_bf16_buckets_and_sparse_rows(tmp_path):
    """The two byte reductions of the gradient exchange against the dense fp32 all-reduce, on the REAL two-rank step
    (HIP cast / pack / scatter kernels, segmented graphs): fp32 + row-sparse source table == dense fp32 up to fp32
    summation order; bf16 buckets + bf16 rows stay within bf16 rounding of it; replicas bit-identical in every mode;
    the row payload replaces the dense table in the byte count (utils/parallel.py:142-181)."""
    out = str(tmp_path)
    modes = [("fp32", "0"), ("fp32", "1"), ("bf16", "1")]
    for m in modes:
        _run(2, out, True, exchange=m)
    res = {m: [np.load(os.path.join(out, "r%d_1_%s_%s.npz" % (r, m[0], m[1]))) for r in range(2)] for m in modes}
    for m in modes:
        assert np.array_equal(res[m][0]["master"], res[m][1]["master"]), m       # replicas identical
        assert np.array_equal(res[m][0]["grad"], res[m][1]["grad"]), m
        assert res[m][0]["exchange"][0] == m[0]
        assert res[m][0]["exchange"][1] == ("src_embedding" if m[1] == "1" else "")
    dense, sp32, sp16 = (res[m][0] for m in modes)
    # fp32 rows in rank order vs ring order of the dense all-reduce: two addends -> identical
    assert np.array_equal(dense["grad"], sp32["grad"]) and np.array_equal(dense["master"], sp32["master"])
    # bf16 exchange, step 1 (same weights in both runs): every element within bf16 rounding of the fp32 sum -- each
    # rank's contribution rounded once (2^-9 relative) and the sum rounded once more
    gd, gb = dense["grad1"], sp16["grad1"]
    assert np.array_equal(dense["grad1"], sp32["grad1"])
    # (the two ranks' contributions may cancel, so the bound is against the largest gradient, not element by element)
    err = np.abs(gd - gb)
    assert err.max() <= 2.0 ** -7 * np.abs(gd).max(), (float(err.max()), float(np.abs(gd).max()))
    assert np.linalg.norm(gd - gb) <= 1e-2 * np.linalg.norm(gd)
    assert abs(np.linalg.norm(gb) - np.linalg.norm(gd)) <= 2e-3 * np.linalg.norm(gd)
    assert sp16["loss"][0] == dense["loss"][0]                 # the forward of step 1 does not see the exchange
    assert np.allclose(sp16["loss"], dense["loss"], rtol=2e-2)  # later steps: Adam amplifies the rounding of tiny gradients
    numel = int(dense["exchange"][3])
    assert int(dense["exchange"][2]) == numel * 4
    assert int(sp32["exchange"][2]) < int(dense["exchange"][2]) and int(sp16["exchange"][2]) < int(sp32["exchange"][2])


def test_bench_contract_with_two_ranks_on_one_gpu():
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one rank per process),
    with the gloo backend and both ranks on cuda:0: one JSON line from rank 0 with the N-rank fields."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, ZERO_DIST_BACKEND="gloo", ZERO_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch_tokens"] == 2 * 64 * 128
    assert out["value"] > 0 and np.isfinite(out["loss"]) and not out["update_skipped"]
    assert "roofline" in out and "cpu_baseline" not in out          # the CPU baseline is a 1-rank field
    assert out["rccl"]["ranks"] == 2 and out["rccl"]["transport"] == "torch.distributed:gloo"
    assert out["rccl"]["ms_per_step_without_exchange"] > 0 and "exposed_allreduce_ms" in out["rccl"]
    # one timed leg per exchange mode in the same process group; the reference-exact (fp32) legs first; the direct
    # transport cannot come up with two ranks on one device and must be reported as skipped, not hang
    legs = out["rccl"]["legs"]
    names = [l["leg"] for l in legs]
    assert names[:4] == ["fp32/torch/dense", "fp32/torch/rows", "bf16/torch/rows", "bf16/torch/dense"], names
    assert all(l["ms_per_step"] > 0 and "exposed_ms" in l for l in legs[:4])
    assert legs[1]["sparse_rows_exchange"] == ["src_embedding"] and legs[0]["sparse_rows_exchange"] == []
    assert legs[1]["bytes_per_rank_per_step"] < legs[0]["bytes_per_rank_per_step"]
    assert legs[2]["bytes_per_rank_per_step"] < legs[1]["bytes_per_rank_per_step"]
    assert [l.get("skipped") is not None for l in legs[4:]] == [True, True], legs[4:]
    assert out["rccl"]["headline_leg"] in names[:4] and out["rccl"]["aborted_leg"] is None
    head = [l for l in legs if l["leg"] == out["rccl"]["headline_leg"]][0]
    assert abs(head["ms_per_step"] - out["ms_per_step"]) < 1e-9
    # fp32 unless bf16 wins by more than 3 %
    best32 = min(l["ms_per_step"] for l in legs[:2])
    assert head["reference_exact"] or head["ms_per_step"] < 0.97 * best32
    assert out["rccl"]["bucket_dtype"] == head["bucket_dtype"]
    assert out["rccl"]["ranks_seen"]["distinct_devices"] == 1 and len(out["rccl"]["ranks_seen"]["device_uuids"]) == 2
    assert "Trainer.step" in out["config"]["timed_loop"]


def test_bench_spawns_its_own_ranks_when_not_under_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the way the driver calls --gpus 1): bench.py starts the
    ranks itself and still prints ONE JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=root, ZERO_DIST_BACKEND="gloo", ZERO_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl"]["ranks"] == 2 and out["value"] > 0


def test_rccl_communicator_through_the_c_abi_single_rank():
    """zk_comm_* (include/zero_hip.h; utils/parallel.py:134-208 -> RCCL): the one-GPU box can only hold a
    one-rank communicator (RCCL refuses two ranks on a device), which still exercises dlopen, the id, init,
    the collectives on the side stream, the event hand-back and destroy; a sum over one rank is the identity."""
    from zero_amd.utils import parallel
    comm = parallel.RcclComm("cuda:0")
    assert comm.world == 1 and comm.lib.raw("zk_comm_size")(comm.handle) == 1
    x = torch.randn(1 << 20, device="cuda:0")
    ref = x.clone()
    ev = comm.all_reduce(x)
    torch.cuda.current_stream().wait_event(ev)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    xb = torch.randn(4099, device="cuda:0").to(torch.bfloat16)
    refb = xb.clone()
    torch.cuda.current_stream().wait_event(comm.all_reduce(xb))
    ids = torch.arange(777, dtype=torch.int32, device="cuda:0")
    out = torch.empty_like(ids)
    torch.cuda.current_stream().wait_event(comm.all_gather(ids, out))
    torch.cuda.synchronize()
    assert torch.equal(xb, refb) and torch.equal(out, ids)
    comm.close()
    comm.close()          # idempotent


def test_flag_handoff_orders_the_communication_stream_like_the_event(monkeypatch):
    """ZERO_HIP_COMM_HANDOFF=flag: the per-bucket ordering of RCCL's stream behind the backward through
    a monotonic device word + a poll kernel instead of an event recorded on the compute stream.  The collective must see
    what the compute stream wrote before the hand-off -- also when the compute stream is still busy for milliseconds
    after the host has enqueued everything -- and give the same bits as the event form."""
    from zero_amd.utils import parallel
    from tests.util_gpu import eng
    e = eng()
    res = {}
    for mode in ("event", "flag"):
        monkeypatch.setenv("ZERO_HIP_COMM_HANDOFF", mode)
        comm = parallel.RcclComm("cuda:0")
        x = torch.zeros(1 << 22, device="cuda:0")
        outs = []
        for it in range(6):
            e.lib.call("zk_spin", 3000, torch.cuda.current_stream().cuda_stream)      # 3 ms of compute-stream work in front
            x.add_(float(it + 1))                                                     # the "gradient" the bucket carries
            ev = comm.all_reduce(x)                                                   # one rank: identity, AFTER the add
            y = torch.empty_like(x)
            with torch.cuda.stream(comm.stream):
                y.copy_(x)                                                            # ordered behind the collective
            outs.append(y)
            torch.cuda.current_stream().wait_event(ev)
        torch.cuda.synchronize()
        assert comm.handoff_errors() == 0
        res[mode] = [float(o[0]) for o in outs] + [float(o[-1]) for o in outs]
        comm.close()
    want = [1.0, 3.0, 6.0, 10.0, 15.0, 21.0]
    assert res["event"] == want + want and res["flag"] == want + want


def test_bench_line_survives_a_leg_that_never_returns():
    """a leg of the optional direct transport that hangs (a collective waiting for a rank that never
    arrives; here: the ZERO_HIP_BENCH_FAKE_HANG test hook) must not cost the run its line.  The watchdog prints it with the
    legs that finished, the headline chosen by the SAME fp32-first rule, `aborted_leg`, and `rccl.ranks_seen` (gathered
    right behind the first leg, before anything optional runs)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, ZERO_DIST_BACKEND="gloo", ZERO_SINGLE_DEVICE="1",
               ZERO_HIP_BENCH_FAKE_HANG="fp32/torch/rows", ZERO_HIP_BENCH_GUARD_S="4")
    # (the SECOND leg hangs: one leg of gloo all-reduces over the 308-MB gradient on the host is all the test pays for)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads(lines[0])
    rc = out["rccl"]
    assert rc["aborted_leg"] == "fp32/torch/rows"
    names = [l["leg"] for l in rc["legs"]]
    assert names == ["fp32/torch/dense"], names
    assert rc["ranks_seen"]["distinct_devices"] == 1 and len(rc["ranks_seen"]["device_uuids"]) == 2
    head = rc["legs"][0]
    assert rc["headline_leg"] == "fp32/torch/dense" and head["reference_exact"]
    assert abs(head["ms_per_step"] - out["ms_per_step"]) < 1e-9 and out["n_gpus"] == 2 and out["value"] > 0
