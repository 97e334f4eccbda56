"""The fp32 decode mode (round 5; decode_dtype = "float32", zero_amd/csrc/zk_f32.hip, zero_amd/models/_decode_f32.py).

Kernels against plain torch fp32 / fp64 references of the same op (tolerances: fp32 round-off of a K-long sum), then the
whole search against the fp32 oracle at toy sizes WITHOUT sharpening the random model: what the bf16 path needs a sharpened
model for (tests/test_gpu_model.py::test_beam_search_token_ids) the fp32 path must deliver as is -- every hypothesis of
every beam token-exact, scores to 1e-5 relative.  The BASELINE-size comparison is tests/test_gpu_fullsize.py."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_torch as rt  # noqa: E402
from tests.common import make_hp, make_batch, perturb  # noqa: E402
from tests.util_gpu import eng  # noqa: E402
from zero_amd.models import model as registry, load_all  # noqa: E402
from zero_amd.models._factory import get_core, reset_cores  # noqa: E402

load_all()
F32 = torch.float32


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("M,N,K,tb", [(128, 512, 512, 0), (128, 512, 2048, 0), (128, 1024, 1024, 0), (3232, 512, 512, 0),
                                      (3232, 1536, 512, 0), (128, 32000, 512, 1), (5, 104, 128, 1), (33, 70, 36, 0),
                                      (1, 512, 512, 0), (700, 2048, 512, 0), (97, 31, 260, 1), (200, 4100, 96, 1)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_f32(M, N, K, tb, act):
    """zk_f32_gemm against an fp64 product: relative error of a K-long fp32 fmaf chain (<= K * 2^-24 of sum |a b|)."""
    e = eng()
    A = _rand(M, K, seed=1)
    B = _rand(N, K, seed=2) if tb else _rand(K, N, seed=2)
    bias = _rand(N, seed=3)
    C = torch.full((M, N + 3), 7.0, device="cuda")          # ldc > N: the padding must stay untouched
    e.lib.call("zk_f32_gemm", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K if tb else N, N + 3, tb, bias.data_ptr(),
               act, e.stream)
    torch.cuda.synchronize()
    ref = A.double() @ (B.double().t() if tb else B.double()) + bias.double()
    if act:
        ref = ref.clamp_min(0)
    mag = A.double().abs() @ (B.double().abs().t() if tb else B.double().abs()) + bias.double().abs()
    err = ((C[:, :N].double() - ref).abs() / mag).max().item()
    assert err < 1e-6, err          # (measured <= 4.2e-7: a few 2^-24 of the sum of magnitudes)
    assert bool((C[:, N:] == 7.0).all())


def test_add_ln_f32_and_embed():
    e = eng()
    rows, H = 131, 512
    x, y = _rand(rows, H, seed=1, scale=3.0) + 5.0, _rand(rows, H, seed=2)
    gam, bet = 1.0 + 0.2 * _rand(H, seed=3), 0.1 * _rand(H, seed=4)
    out = torch.empty(rows, H, device="cuda")
    e.lib.call("zk_f32_add_ln", x.data_ptr(), y.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), rows, H, 1e-8,
               e.stream)
    s = (x + y).double()
    mean = s.mean(-1, keepdim=True)
    var = ((s - mean) ** 2).mean(-1, keepdim=True)
    ref = gam.double() * (s - mean) * torch.rsqrt(var + 1e-8) + bet.double()
    torch.cuda.synchronize()
    assert (out.double() - ref).abs().max().item() < 3e-6
    # no residual operand; constant row -> offset (func.py:289-303 with eps inside the root)
    c = torch.full((4, H), 2.5, device="cuda")
    e.lib.call("zk_f32_add_ln", c.data_ptr(), None, gam.data_ptr(), bet.data_ptr(), out.data_ptr(), 4, H, 1e-8, e.stream)
    torch.cuda.synchronize()
    assert torch.allclose(out[:4], bet.expand(4, H), atol=1e-6)
    # embedding x sqrt(H) + bias + timing; all-pad flag zeroes the embedding part only
    V, L, B = 50, 7, 3
    tab, bias = _rand(V, H, seed=5), _rand(H, seed=6)
    tim = e.timing(L + 1, H)
    ids = torch.randint(0, V, (B, L), dtype=torch.int32, device="cuda")
    o2 = torch.empty(B * L, H, device="cuda")
    e.lib.call("zk_f32_embed", ids.data_ptr(), B * L, L, tab.data_ptr(), bias.data_ptr(), tim.data_ptr(), int(tim.shape[0]),
               o2.data_ptr(), H, float(H) ** 0.5, 0, None, None, e.stream)
    ref2 = tab[ids.long().view(-1)] * (H ** 0.5) + bias + tim[:L].repeat(B, 1)
    torch.cuda.synchronize()
    assert torch.allclose(o2, ref2, rtol=2e-7, atol=2e-6)
    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    pos = torch.tensor([5], dtype=torch.int32, device="cuda")
    e.lib.call("zk_f32_embed", ids.data_ptr(), B, 1, tab.data_ptr(), bias.data_ptr(), tim.data_ptr(), int(tim.shape[0]),
               o2.data_ptr(), H, float(H) ** 0.5, 0, pos.data_ptr(), flag.data_ptr(), e.stream)
    torch.cuda.synchronize()
    assert torch.equal(o2[:B], tim[5:6].repeat(B, 1))


@pytest.mark.parametrize("B,Lq,Lk,group,masked,cached", [(3, 9, 9, 1, True, False), (8, 1, 23, 4, True, False),
                                                        (8, 1, 40, 1, False, True), (2, 101, 101, 1, True, False)])
def test_attention_f32(B, Lq, Lk, group, masked, cached):
    """zk_f32_attn against func.py:218-256 in fp64: q pre-scaled, finite -1e8 mask (a fully masked row is uniform),
    beam rows sharing a sentence's keys (kv_group), and the cached form (only *nkeys_dev + 1 keys exist)."""
    e = eng()
    nh, d = 8, 64
    H = nh * d
    Bk = B // group
    q = _rand(B * Lq, H, seed=1)
    k, v = _rand(Bk * Lk, H, seed=2), _rand(Bk * Lk, H, seed=3)
    mask = torch.ones(Bk, Lk, device="cuda")
    if masked:
        mask[0, Lk // 2:] = 0.0
        if Bk > 1:
            mask[1, :] = 0.0            # fully masked: uniform weights, not NaN
    nk = torch.tensor([Lk // 3], dtype=torch.int32, device="cuda")
    out = torch.empty(B * Lq, H, device="cuda")
    e.lib.call("zk_f32_attn", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, nh, Lq, Lk, d, H, H, H, H,
               Lq * H, Lk * H, Lk * H, Lq * H, mask.data_ptr() if masked else None, Lk, group, d ** -0.5, 1e8,
               nk.data_ptr() if cached else None, None, None, 0, 0, None, e.stream)
    torch.cuda.synchronize()
    n = Lk // 3 + 1 if cached else Lk
    qd = (q.view(B, Lq, nh, d).permute(0, 2, 1, 3) * (d ** -0.5)).double()
    kd = k.view(Bk, Lk, nh, d).permute(0, 2, 1, 3).double().repeat_interleave(group, 0)[:, :, :n]
    vd = v.view(Bk, Lk, nh, d).permute(0, 2, 1, 3).double().repeat_interleave(group, 0)[:, :, :n]
    lg = qd @ kd.transpose(-1, -2)
    if masked:
        lg = (lg.float() + ((1.0 - mask.repeat_interleave(group, 0))[:, None, None, :n] * -1e8)).double()
    ref = (torch.softmax(lg, -1) @ vd).permute(0, 2, 1, 3).reshape(B * Lq, H)
    assert (out.double() - ref).abs().max().item() < 5e-6


def test_aan_step_and_gate_f32():
    e = eng()
    rows, H = 12, 256
    x, cache = _rand(rows, H, seed=1), _rand(rows, H, seed=2)
    c0 = cache.clone()
    cat = torch.empty(rows, 2 * H, device="cuda")
    t = torch.tensor([6], dtype=torch.int32, device="cuda")
    e.lib.call("zk_f32_aan_step", x.data_ptr(), cache.data_ptr(), cat.data_ptr(), rows, H, 0, t.data_ptr(), e.stream)
    torch.cuda.synchronize()
    # (true division, as TF's realdiv and torch-CPU do it; torch on the GPU would multiply by the reciprocal of a scalar)
    want = torch.from_numpy(((x + c0).cpu().numpy() / np.float32(7.0)).astype(np.float32)).cuda()
    assert torch.equal(cache, x + c0) and torch.equal(cat[:, :H], x) and torch.equal(cat[:, H:], want)
    z = _rand(rows, 2 * H, seed=3, scale=2.0)
    g = torch.empty(rows, H, device="cuda")
    e.lib.call("zk_f32_gate", z.data_ptr(), cat.data_ptr(), g.data_ptr(), rows, H, e.stream)
    torch.cuda.synchronize()
    ref = torch.sigmoid(z[:, :H]) * cat[:, :H] + torch.sigmoid(z[:, H:]) * cat[:, H:]
    assert torch.allclose(g, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M,N,K,tb", [(128, 512, 512, 0), (128, 512, 2048, 0), (128, 1024, 1024, 0), (128, 2048, 512, 0),
                                      (1280, 1536, 512, 0), (128, 32000, 512, 1), (97, 31, 260, 1), (640, 512, 2048, 0)])
def test_gemm_f32_k_sliced_against_the_round5_kernel(M, N, K, tb):
    """Round 6: the K-sliced kernel (every operand chunk requested up front, 4 / 8 / 16 waves per tile) and the column-major
    tile order of the logits product against the round-5 kernels (zk_f32_gemm_legacy): the same products summed in a
    different association -- equal to a few units of fp32 round-off of the sum of magnitudes, both equally far from fp64."""
    e = eng()
    A = _rand(M, K, seed=1)
    B = _rand(N, K, seed=2) if tb else _rand(K, N, seed=2)
    bias = _rand(N, seed=3)
    out = []
    for legacy in (1, 2, 0):         # round-5 kernels; round 6 without the 16 x 16 tiles; the default
        C = torch.full((M, N), 7.0, device="cuda")
        e.lib.raw("zk_f32_gemm_legacy")(legacy)
        try:
            e.lib.call("zk_f32_gemm", A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K if tb else N, N, tb,
                       bias.data_ptr(), 1, e.stream)
            torch.cuda.synchronize()
        finally:
            e.lib.raw("zk_f32_gemm_legacy")(0)
        out.append(C)
    ref = (A.double() @ (B.double().t() if tb else B.double()) + bias.double()).clamp_min(0)
    mag = A.double().abs() @ (B.double().abs().t() if tb else B.double().abs()) + bias.double().abs()
    errs = [((o_.double() - ref).abs() / mag).max().item() for o_ in out]
    assert max(errs) < 1e-6, errs
    for o_ in out[1:]:
        assert ((out[0] - o_).double().abs() / mag).max().item() < 1e-6
    if tb and N >= 2048:
        # the LDS-staged logits kernel keeps the round-5 kernel's fmaf chain per output: the same bits
        assert torch.equal(out[0], out[2])


def test_ln_fused_f32_equals_the_launches_it_replaces():
    """zk_f32_ln_fused / zk_f32_embed_step (round 6) against zk_f32_gate + zk_f32_add_ln + zk_f32_aan_step /
    zk_all_equal + zk_f32_embed: the same values up to the association of the two row sums (mean / variance)."""
    e = eng()
    rows, H = 131, 512
    x, y = _rand(rows, H, seed=1, scale=3.0) + 5.0, _rand(rows, H, seed=2)
    gam, bet = 1.0 + 0.2 * _rand(H, seed=3), 0.1 * _rand(H, seed=4)
    t = torch.tensor([6], dtype=torch.int32, device="cuda")
    # plain + next layer's average attention
    want = torch.empty(rows, H, device="cuda")
    e.lib.call("zk_f32_add_ln", x.data_ptr(), y.data_ptr(), gam.data_ptr(), bet.data_ptr(), want.data_ptr(), rows, H, 1e-8, e.stream)
    cache0 = _rand(rows, H, seed=5)
    cache_w, cat_w = cache0.clone(), torch.empty(rows, 2 * H, device="cuda")
    e.lib.call("zk_f32_aan_step", want.data_ptr(), cache_w.data_ptr(), cat_w.data_ptr(), rows, H, 0, t.data_ptr(), e.stream)
    got, cache_g, cat_g = torch.empty_like(want), cache0.clone(), torch.empty_like(cat_w)
    e.lib.call("zk_f32_ln_fused", x.data_ptr(), y.data_ptr(), None, None, gam.data_ptr(), bet.data_ptr(), got.data_ptr(), rows, H,
               1e-8, cache_g.data_ptr(), cat_g.data_ptr(), 0, t.data_ptr(), e.stream)
    torch.cuda.synchronize()
    assert (got - want).abs().max().item() < 2e-6
    assert torch.equal(cat_g[:, :H], got) and torch.equal(cache_g, got + cache0)
    assert torch.equal(cat_g[:, H:], torch.from_numpy(((got + cache0).cpu().numpy() / np.float32(7.0)).astype(np.float32)).cuda())
    # the gate as the producer of y; the residual is cat[:, :H]
    z, cat = _rand(rows, 2 * H, seed=6, scale=2.0), _rand(rows, 2 * H, seed=7)
    g = torch.empty(rows, H, device="cuda")
    e.lib.call("zk_f32_gate", z.data_ptr(), cat.data_ptr(), g.data_ptr(), rows, H, e.stream)
    xc = cat[:, :H].contiguous()
    e.lib.call("zk_f32_add_ln", xc.data_ptr(), g.data_ptr(), gam.data_ptr(), bet.data_ptr(), want.data_ptr(), rows, H, 1e-8, e.stream)
    e.lib.call("zk_f32_ln_fused", None, None, z.data_ptr(), cat.data_ptr(), gam.data_ptr(), bet.data_ptr(), got.data_ptr(), rows, H,
               1e-8, None, None, 0, None, e.stream)
    torch.cuda.synchronize()
    assert (got - want).abs().max().item() < 3e-6
    # decoder input of one position: all-pad test inside the launch, first layer's average attention
    V = 50
    tab, bias = _rand(V, H, seed=8), _rand(H, seed=9)
    tim = e.timing(12, H)
    for ids in (torch.randint(1, V, (rows,), dtype=torch.int32, device="cuda"), torch.zeros(rows, dtype=torch.int32, device="cuda")):
        flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        e.lib.call("zk_all_equal", ids.data_ptr(), rows, 0, flag.data_ptr(), e.stream)
        e.lib.call("zk_f32_embed", ids.data_ptr(), rows, 1, tab.data_ptr(), bias.data_ptr(), tim.data_ptr(), int(tim.shape[0]),
                   want.data_ptr(), H, float(H) ** 0.5, 0, t.data_ptr(), flag.data_ptr(), e.stream)
        cache_w = cache0.clone()
        e.lib.call("zk_f32_aan_step", want.data_ptr(), cache_w.data_ptr(), cat_w.data_ptr(), rows, H, 0, t.data_ptr(), e.stream)
        cache_g = cache0.clone()
        e.lib.call("zk_f32_embed_step", ids.data_ptr(), rows, tab.data_ptr(), bias.data_ptr(), tim.data_ptr(), int(tim.shape[0]),
                   got.data_ptr(), H, float(H) ** 0.5, 0, t.data_ptr(), 0, cache_g.data_ptr(), cat_g.data_ptr(), e.stream)
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(cache_g, cache_w) and torch.equal(cat_g, cat_w)
    assert torch.equal(got, tim[6:7].repeat(rows, 1))          # all ids = pad: timing signal only


@pytest.mark.parametrize("model", ["transformer_aan", "transformer"])
def test_fp32_step_with_folded_launches_equals_one_launch_per_op(model, monkeypatch):
    """ZERO_HIP_F32_FUSE=0 (one launch per op, round 5) and the default step (round 6) decode the same hypotheses with the
    same scores to fp32 round-off."""
    from zero_amd.main import tower_infer_graph
    rng = np.random.default_rng(31)
    hp = make_hp(model, decode_dtype="float32")
    Pn = perturb(rt.init_params(hp, model, seed=32), rng)
    src, _ = make_batch(rng, 6, 11, 5, hp.src_vocab.size(), hp.tgt_vocab.size())
    hp = copy.copy(hp)
    hp.beam_size = 4
    hp.search_mode = "cache"
    res = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("ZERO_HIP_F32_FUSE", fuse)
        reset_cores()
        get_core(hp, model, Pn)
        seqs, scores = tower_infer_graph({"source": src}, registry.get_model(model), hp)
        res.append((np.asarray(seqs).copy(), np.asarray(scores).copy()))
    assert np.array_equal(res[0][0], res[1][0])
    fin = res[0][1] > -1e30
    assert np.allclose(res[0][1][fin], res[1][1][fin], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("model,kw", [("transformer_aan", {}), ("transformer", {}), ("transformer_aan", {"use_ffn": True}),
                                      ("transformer_rpr", {}), ("transformer_fuse", {})])
@pytest.mark.parametrize("K", [1, 4])
def test_fp32_decode_is_token_exact_without_sharpening(model, kw, K):
    """The whole search in the fp32 mode against the fp32 oracle on an UNSHARPENED random toy model: every hypothesis of
    every beam identical, scores within 1e-5 relative, the same number of decode steps; the step graphs are replayed
    (device-resident bookkeeping) and give what the eager host-bookkeeping path gives."""
    from zero_amd.main import tower_infer_graph
    reset_cores()
    rng = np.random.default_rng(21)
    hp = make_hp(model, decode_dtype="float32", **kw)
    Pn = perturb(rt.init_params(hp, model, seed=22), rng)
    src, _ = make_batch(rng, 7, 12, 5, hp.src_vocab.size(), hp.tgt_vocab.size())
    hp = copy.copy(hp)
    hp.beam_size = K
    hp.search_mode = "cache"
    enc, dec = rt.infer_fn(hp, rt.to_torch(Pn), model)
    ref = rt.beam_search({"source": torch.tensor(src)}, enc, dec, hp)
    get_core(hp, model, Pn)
    seqs, scores = tower_infer_graph({"source": src}, registry.get_model(model), hp)
    L = min(seqs.shape[2], ref["seq"].shape[2])
    assert np.array_equal(np.asarray(seqs)[:, :, :L], ref["seq"][:, :, :L]), (seqs, ref["seq"])
    assert not np.asarray(seqs)[:, :, L:].any() and not ref["seq"][:, :, L:].any()
    fin = ref["score"] > -1e30
    assert np.allclose(np.asarray(scores)[fin], ref["score"][fin], rtol=1e-5, atol=1e-6)
    # the traced host-bookkeeping path (the one the full-size test reads its candidate tables from) agrees
    hp.search_trace = []
    seqs2, scores2 = tower_infer_graph({"source": src}, registry.get_model(model), hp)
    assert np.array_equal(np.asarray(seqs2), np.asarray(seqs)) and len(hp.search_trace) > 0


def test_fp32_decode_refuses_what_it_does_not_cover():
    """search_mode = "dev" re-runs the bf16 training-path decoder (transformer.py:277-281): the fp32 mode says so."""
    from zero_amd.hip import ZeroHipError
    from zero_amd.main import tower_infer_graph
    reset_cores()
    rng = np.random.default_rng(2)
    hp = make_hp("transformer", decode_dtype="float32")
    Pn = perturb(rt.init_params(hp, "transformer", seed=2), rng)
    src, _ = make_batch(rng, 3, 6, 5, hp.src_vocab.size(), hp.tgt_vocab.size())
    hp.search_mode = "dev"
    get_core(hp, "transformer", Pn)
    with pytest.raises(ZeroHipError):
        tower_infer_graph({"source": src}, registry.get_model("transformer"), hp)


def test_attention_f32_relative_positions():
    """modules/rpr.py:10-75 inside zk_f32_attn: logits += q . r_k[clip(i - j) + m], o += sum_j p_j r_v[...], for a full
    (encoder) block and for one cached decode row at position *q_pos_dev."""
    e = eng()
    nh, d, m = 2, 64, 4
    H = nh * d
    for (B, Lq, Lk, pos) in ((3, 11, 11, None), (4, 1, 13, 7)):
        q, k, v = _rand(B * Lq, H, seed=1), _rand(B * Lk, H, seed=2), _rand(B * Lk, H, seed=3)
        rk, rv = _rand(2 * m + 1, d, seed=4), _rand(2 * m + 1, d, seed=5)
        out = torch.empty(B * Lq, H, device="cuda")
        pd = torch.tensor([pos or 0], dtype=torch.int32, device="cuda")
        e.lib.call("zk_f32_attn", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, nh, Lq, Lk, d, H, H, H, H,
                   Lq * H, Lk * H, Lk * H, Lq * H, None, 0, 1, d ** -0.5, 1e8, pd.data_ptr() if pos is not None else None,
                   rk.data_ptr(), rv.data_ptr(), m, 0, pd.data_ptr() if pos is not None else None, e.stream)
        torch.cuda.synchronize()
        n = pos + 1 if pos is not None else Lk
        qd = (q.view(B, Lq, nh, d).permute(0, 2, 1, 3) * (d ** -0.5)).double()
        kd = k.view(B, Lk, nh, d).permute(0, 2, 1, 3).double()[:, :, :n]
        vd = v.view(B, Lk, nh, d).permute(0, 2, 1, 3).double()[:, :, :n]
        i = (torch.arange(Lq) + (pos or 0))[:, None]
        idx = (torch.clamp(i - torch.arange(n)[None, :], -m, m) + m).cuda()
        lg = qd @ kd.transpose(-1, -2) + torch.einsum("bhqd,qkd->bhqk", qd, rk.double()[idx])
        pr = torch.softmax(lg, -1)
        ref = (pr @ vd + torch.einsum("bhqk,qkd->bhqd", pr, rv.double()[idx])).permute(0, 2, 1, 3).reshape(B * Lq, H)
        assert (out.double() - ref).abs().max().item() < 5e-6
