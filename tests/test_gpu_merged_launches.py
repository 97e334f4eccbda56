"""Round 6: small launches of the training step merged pairwise -- both input embeddings
(zk_embed_fwd_pair), the two embedding-gradient scatters (zk_embed_bwd_sorted_pair), the two bias column sums
(zk_colsum_pair), per-sentence loss + mean (k_loss_tail).  Each merged launch against the launches it replaces, then the
whole training step with ZERO_HIP_MERGE_SMALL = 0 / 1 (transformer.py:16-33, 88-119, 198-211 and their gradients)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util_gpu import eng, rand_bf, mat  # noqa: E402

F32 = torch.float32


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_embed_fwd_pair_equals_the_two_launches(drop):
    e = eng()
    e.set_seed(3)
    B, La, Lb, H, Va, Vb = 5, 9, 7, 128, 61, 47
    ta, tb = rand_bf(Va, H, seed=1), rand_bf(Vb, H, seed=2)
    bias = torch.randn(H, device="cuda")
    g = torch.Generator().manual_seed(5)
    ia = torch.randint(0, Va, (B, La), generator=g, dtype=torch.int32).cuda()
    ib = torch.randint(0, Vb, (B, Lb), generator=g, dtype=torch.int32).cuda()
    wa, wb = torch.empty(B * La, H, dtype=torch.bfloat16, device="cuda"), torch.empty(B * Lb, H, dtype=torch.bfloat16, device="cuda")
    e.embed_fwd(ia, ta, bias, mat(wa), B, La, H, drop_p=drop, sid=9001)
    e.embed_fwd(ib, tb, bias, mat(wb), B, Lb, H, shift=True, drop_p=drop, sid=9002)
    ga, gb = torch.zeros_like(wa), torch.zeros_like(wb)
    e.embed_fwd_pair(ia, ta, mat(ga), La, 9001, ib, tb, mat(gb), Lb, 9002, bias, B, H, drop_p=drop)
    torch.cuda.synchronize()
    assert torch.equal(ga, wa) and torch.equal(gb, wb)


def _sorted(ids, V):
    """rows grouped by id (stable), as zk_batch_prep leaves them"""
    flat = ids.reshape(-1).cpu().numpy()
    order = np.argsort(flat, kind="stable").astype(np.int32)
    uid, start = np.unique(flat[order], return_index=True)
    seg = np.concatenate([start, [len(flat)]]).astype(np.int32)
    dev = lambda a: torch.tensor(a, dtype=torch.int32, device="cuda")
    pad = np.zeros(len(flat) + 1, np.int32)
    pad[:len(seg)] = seg
    u = np.zeros(len(flat), np.int32)
    u[:len(uid)] = uid
    return {"rows": dev(order), "seg": dev(pad), "uid": dev(u), "n": dev([len(uid)]), "max_uniq": len(flat)}


@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_embed_bwd_pair_and_colsum_pair_equal_the_launches_they_replace(drop):
    e = eng()
    e.set_seed(4)
    B, La, Lb, H, Va, Vb = 6, 11, 8, 128, 40, 33
    g = torch.Generator().manual_seed(6)
    ia = torch.randint(0, Va, (B, La), generator=g, dtype=torch.int32).cuda()
    ib = torch.randint(0, Vb, (B, Lb), generator=g, dtype=torch.int32).cuda()
    da, db = rand_bf(B * La, H, seed=3), rand_bf(B * Lb, H, seed=4)
    sa, sb = _sorted(ia, Va), _sorted(ib, Vb)
    base_b = torch.randn(Vb, H, device="cuda")              # side b accumulates onto what a weight-gradient GEMM left there
    wa, wb = torch.zeros(Va, H, device="cuda"), base_b.clone()
    e.embed_bwd_sorted(sb, mat(db), wb, H, accumulate=True, drop_p=drop, sid=9002)
    e.embed_bwd_sorted(sa, mat(da), wa, H, accumulate=False, drop_p=drop, sid=9001)
    ga, gb = torch.zeros(Va, H, device="cuda"), base_b.clone()
    e.embed_bwd_sorted_pair(sb, mat(db), gb, True, 9002, sa, mat(da), ga, False, 9001, H, drop_p=drop)
    torch.cuda.synchronize()
    touched_a = torch.unique(ia.long().view(-1))
    assert torch.equal(ga[touched_a], wa[touched_a]) and torch.equal(gb, wb)
    # bias gradient: colsum(db without the rows r % Lb == 0) + colsum(da)
    want, got = torch.empty(H, device="cuda"), torch.empty(H, device="cuda")
    e.colsum(mat(db), want, skip_L=Lb, accumulate=False, drop_p=drop, sid=9002)
    e.colsum(mat(da), want, skip_L=0, accumulate=True, drop_p=drop, sid=9001)
    e.colsum_pair(mat(db), Lb, 9002, mat(da), 0, 9001, got, drop_p=drop)
    torch.cuda.synchronize()
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    if drop == 0.0:
        keep = torch.ones(B * Lb, dtype=torch.bool, device="cuda")
        keep[::Lb] = False
        ref = db.float()[keep].sum(0) + da.float().sum(0)
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-3)


def test_loss_tail_equals_per_sample_then_mean():
    e = eng()
    for B, L in ((64, 64), (7, 13), (300, 130), (1, 5)):
        g = torch.Generator().manual_seed(B)
        ids = torch.randint(0, 4, (B, L), generator=g, dtype=torch.int32)
        ids[:, 0] = 3                                   # no empty sentence
        ids = ids.cuda()
        ce = torch.rand(B, L, generator=g).cuda()
        res = []
        for two in (1, 0):
            old = e.lib.raw("zk_tune")(16, two)
            try:
                ps, loss = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda")
                e.loss_reduce(ce, ids, ps, loss, B, L)
                torch.cuda.synchronize()
            finally:
                e.lib.raw("zk_tune")(16, old)
            res.append((ps, loss))
        mk = (ids != 0).float()
        ref = (ce * mk).sum(1) / mk.sum(1)
        assert torch.allclose(res[1][0], ref, rtol=1e-5) and torch.allclose(res[1][1], ref.mean(), rtol=1e-5)
        if L <= 64:
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        else:
            assert torch.allclose(res[0][0], res[1][0], rtol=1e-6) and torch.allclose(res[0][1], res[1][1], rtol=1e-6)


@pytest.mark.parametrize("model", ["transformer", "transformer_aan"])
def test_training_steps_with_merged_launches(model, monkeypatch):
    """The captured training step with the merged launches against the round-5 launches: same losses (the per-sentence
    loss and the mean are the same bits), weights equal to fp32 round-off of the bias gradient's summation order; five
    launches fewer per step."""
    from tests.test_gpu_model import _setup
    from zero_amd.main import Trainer
    from zero_amd.models._factory import reset_cores
    hp, Pn, src, tgt = _setup(model)
    runs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("ZERO_HIP_MERGE_SMALL", flag)
        reset_cores()
        tr = Trainer(hp, initializer=Pn)
        tr.prepare_static({"source": src, "target": tgt})
        losses = [float(tr.step_static(use_graph=True).cpu()[0]) for _ in range(4)]
        torch.cuda.synchronize()
        runs[flag] = (losses, tr.store.export("master"), tr.core.eng.last_graph_nodes)
    assert np.allclose(runs["0"][0], runs["1"][0], rtol=2e-6, atol=0), (runs["0"][0], runs["1"][0])
    for k, a in runs["0"][1].items():
        b = runs["1"][1][k]
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(a).max()), k
    assert runs["1"][2] <= runs["0"][2] - 6, (runs["0"][2], runs["1"][2])
