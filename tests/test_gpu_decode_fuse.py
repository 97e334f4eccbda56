"""Fused attention sub-layers of a cached decode step (zk_dec_cross / zk_dec_self + zk_ln_decode with partial sums,
zero_amd/csrc/zk_decfuse.hip) against the launch-per-op path of the same step (transformer.py:120-175 at Lq = 1,
func.py:124-287).  The two paths round to bf16 at the same points and differ only in the order of the fp32 sums, so
  * logits of a step agree to a few bf16 ulps of the activations feeding them (tolerance below, measured ~3e-3),
  * beam search returns the same token ids (random model with a sharpened output distribution) and scores within 2e-2;
the oracle comparison of the fused path itself is tests/test_gpu_model.py::test_beam_search_token_ids and
tests/test_gpu_fullsize.py (both run the default = fused path)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ref_torch as rt  # noqa: E402
from tests.common import make_hp, make_batch, perturb  # noqa: E402
from zero_amd.models._factory import get_core, reset_cores  # noqa: E402
from zero_amd.models import model as registry, load_all  # noqa: E402

load_all()

SIZES = {"small": dict(H=128, F=256, heads=2, layers=2), "base": dict(H=512, F=1024, heads=8, layers=2),
         "wide": dict(H=1024, F=1024, heads=16, layers=1)}


def _model(model, size, seed, K):
    reset_cores()
    rng = np.random.default_rng(seed)
    hp = make_hp(model, beam_size=K, **SIZES[size])
    hp.search_mode = "cache"
    Pn = perturb(rt.init_params(hp, model, seed=seed + 1), rng)
    Pn["tgt_embedding"] = (Pn["tgt_embedding"] * 6.0).astype(np.float32)
    src, _ = make_batch(rng, 5, 9, 11, hp.src_vocab.size(), hp.tgt_vocab.size())
    return hp, Pn, src


@pytest.mark.parametrize("model", ["transformer", "transformer_aan", "transformer_rpr"])
@pytest.mark.parametrize("size,K", [("small", 1), ("small", 4), ("base", 4), ("base", 2), ("base", 8), ("wide", 4)])
def test_step_logits_fused_vs_per_op(model, size, K, monkeypatch):
    """Consecutive eager decode steps (time = 0 .. 3, caches carried; with relative positions 0 .. 7, i.e. past the
    clipping distance max_relative_position = 4 of the test model, in both directions of the cross attention) on both
    paths: logits compared.  transformer_rpr (round 4): the relative-position terms of modules/rpr.py:10-75 run inside
    the fused launch instead of the launch-per-op attention."""
    hp, Pn, src = _model(model, size, 21, K)
    V = hp.tgt_vocab.size()
    rng = np.random.default_rng(5)
    nsteps = 8 if model == "transformer_rpr" else 4
    toks = [rng.integers(3, V, size=(src.shape[0] * K,)).astype(np.int32) for _ in range(nsteps)]
    outs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("ZERO_HIP_DECODE_FUSE_ATT", fuse)
        reset_cores()
        core = get_core(hp, model, Pn)
        enc, dec = registry.get_model(model).infer_fn(hp)
        state = enc(src)
        steps = []
        for t, tok in enumerate(toks):
            logits, state = dec(torch.from_numpy(tok).to(core.eng.device), state, t)
            torch.cuda.synchronize()
            steps.append(logits.t[:, :V].float().cpu().numpy().copy())
            # same beams kept on both paths: identity reorder (exercises the ping-pong halves)
            state.reorder(torch.arange(src.shape[0] * K, dtype=torch.int32, device=core.eng.device))
        outs[fuse] = steps
    for t in range(nsteps):
        a, b = outs["0"][t], outs["1"][t]
        scale = np.abs(a).max()
        err = np.abs(a - b).max() / scale
        print("%s %s K=%d step %d: max |logit diff| / max |logit| = %.2e" % (model, size, K, t, err))
        assert np.isfinite(b).all()
        assert err < 2e-2, (t, err)
        assert (a.argmax(1) == b.argmax(1)).mean() >= 0.95


@pytest.mark.parametrize("model", ["transformer", "transformer_aan", "transformer_rpr"])
@pytest.mark.parametrize("size,K", [("small", 4), ("base", 4), ("base", 1)])
def test_beam_search_fused_vs_per_op(model, size, K, monkeypatch):
    """Whole searches (device-resident bookkeeping, replayed step graphs): same hypotheses, scores within 2e-2."""
    from zero_amd import search
    hp, Pn, src = _model(model, size, 8, K)
    outs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("ZERO_HIP_DECODE_FUSE_ATT", fuse)
        reset_cores()
        get_core(hp, model, Pn)
        enc, dec = registry.get_model(model).infer_fn(hp)
        outs[fuse] = search.beam_search({"source": src}, enc, dec, hp)
    assert outs["0"]["steps"] == outs["1"]["steps"] > 2
    same = (outs["0"]["seq"][:, 0] == outs["1"]["seq"][:, 0]).all(axis=-1)
    diff = np.abs(outs["0"]["score"][:, 0] - outs["1"]["score"][:, 0])
    print("%s %s K=%d: %d/%d best hypotheses token-exact, max score diff %.2e" % (model, size, K, same.sum(), len(same),
                                                                             diff.max()))
    assert same.all()
    assert diff.max() < 2e-2


def test_fused_path_is_refused_where_it_does_not_apply(monkeypatch):
    """Shapes outside the fused kernels (head width != 64) keep the launch-per-op path and still decode."""
    from zero_amd import search
    reset_cores()
    rng = np.random.default_rng(3)
    hp = make_hp("transformer", H=96, F=128, heads=3, layers=1, beam_size=2)
    hp.search_mode = "cache"
    Pn = perturb(rt.init_params(hp, "transformer", seed=4), rng)
    src, _ = make_batch(rng, 3, 7, 7, hp.src_vocab.size(), hp.tgt_vocab.size())
    get_core(hp, "transformer", Pn)
    enc, dec = registry.get_model("transformer").infer_fn(hp)
    out = search.beam_search({"source": src}, enc, dec, hp)
    assert out["steps"] > 1


def test_dec_cross_argument_checks():
    from zero_amd import hip
    lib = hip.lib()
    x = torch.zeros(8, 192, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(192, 192, dtype=torch.bfloat16, device="cuda")
    f = torch.zeros(3 * 8 * 192, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(hip.ZeroHipError, match="power of two"):
        lib.call("zk_dec_cross", x.data_ptr(), None, None, None, None, 192, 1e-6, None, None, None, 0, 0, None, None, None,
                 1.0, None, w.data_ptr(), 192, f.data_ptr(), x.data_ptr(), x.data_ptr(), 192, 192, 192 * 4,
                 192 * 4, None, 0, w.data_ptr(), 192, f.data_ptr(), 2, 4, 3, 4, 0.125, 1e9, None, None, 0, 0, None, st)


@pytest.mark.parametrize("M,H,F,splits", [(128, 512, 2048, 4), (64, 512, 2048, 4), (128, 128, 256, 2), (100, 512, 2048, 4)])
def test_feed_forward_pair_in_one_launch(M, H, F, splits):
    """zk_ffn_pair against zk_gemm(act = 1) followed by zk_gemm_parts: h and every partial product bit for bit (same tile
    function, same K order), launch after launch on the same arrival counter (the barrier's window advances by the number
    of phase-1 tiles per launch) with the input changing in between."""
    import ctypes
    from tests.util_gpu import eng, rand_bf, mat
    e = eng()
    if not e.lib.experiments:
        pytest.skip("the feed-forward pair in one launch is an EXPERIMENTS=1 build (measured, slower)")
    W1, W2 = rand_bf(H, F, seed=2, scale=0.05), rand_bf(F, H, seed=3, scale=0.05)
    b1 = (torch.randn(F, generator=torch.Generator().manual_seed(4)) * 0.1).cuda()
    for rep in range(12):
        x = rand_bf(M, H, seed=10 + rep)
        h0 = torch.empty(M, F, dtype=torch.bfloat16, device="cuda")
        p0 = torch.zeros(splits, M, H, device="cuda")
        e.gemm(mat(x), mat(W1), mat(h0), M, F, H, 0, 0, bias=b1, act=1)
        n0 = ctypes.c_int(0)
        e.lib.call("zk_gemm_parts", h0.data_ptr(), W2.data_ptr(), p0.data_ptr(), M, H, F, F, H, 0, 0, splits, ctypes.byref(n0), e.stream)
        h1 = torch.full((M, F), 3.0, dtype=torch.bfloat16, device="cuda")
        p1 = torch.full((splits, M, H), 7.0, device="cuda")
        n1 = e.ffn_pair(mat(x), mat(W1), b1, mat(h1), mat(W2), p1, splits)
        torch.cuda.synchronize()
        assert n1 == n0.value
        assert torch.equal(h1, h0)
        assert torch.equal(p1[:n1], p0[:n1])
    assert e.sync_ln_errors() == 0
