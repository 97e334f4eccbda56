"""Layer programs (zero_amd/csrc/zk_layer.hip): the encoder stack as ONE persistent launch -- the sentences dealt to
the 8 XCDs, a barrier among an XCD's workgroups between ops -- must give bit-identical results to the launch-per-op
path (same tile functions, same arithmetic order), with and without dropout, and must fall back to ordinary launches
for shapes it does not cover."""
import numpy as np
import pytest
import torch

from zero_amd import hip as _hip

# an experiment (negative result): only in a `make EXPERIMENTS=1` library
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _hip.lib().experiments, reason="layer program needs `make EXPERIMENTS=1`")]

from oracle import ref_torch as rt  # noqa: E402
from tests.common import make_hp, perturb  # noqa: E402
from zero_amd.models._factory import get_core, reset_cores  # noqa: E402


def _run(hp, Pn, src, tgt, program, seed=11):
    reset_cores()
    core = get_core(hp, hp.model_name, Pn)
    core.eng.programs_enabled = program
    core.sync_ln_mode = False      # the program's ops are those of the launch-per-op structure of rounds 1-3 (zk_gemm, zk_add_ln_fwd, ..)
    core.eng.set_seed(seed)
    batch = core.upload(src, tgt)
    loss, ps, _ = core.forward(batch, train=True, save=True)
    core.backward()
    torch.cuda.synchronize()
    B, Ls = batch["B"], batch["Ls"]
    last = hp.num_encoder_layer - 1
    enc = core.eng.mat("e%d.ff.o" % last, B * Ls, hp.hidden_size).t.clone()
    mid = core.eng.mat("e0.sa.s", B * Ls, hp.hidden_size).t.clone()
    status = core.eng.program_status() if program else None
    return float(loss.cpu()), enc, mid, core.store.grad.clone(), status, getattr(core.eng, "_prog_cache", None)


@pytest.mark.parametrize("cfg", ["small", "base"])
@pytest.mark.parametrize("drop", [0.0, 0.1])
def test_encoder_program_is_bit_identical(cfg, drop):
    rng = np.random.default_rng(5)
    if cfg == "small":
        hp = make_hp("transformer", H=128, F=256, heads=2, layers=2, Vs=300, Vt=300)
        B, Ls, Lt = 16, 32, 24
    else:
        hp = make_hp("transformer", H=512, F=2048, heads=8, layers=6, Vs=2000, Vt=2000)
        B, Ls, Lt = 64, 64, 64
    hp.override_from_dict(dict(dropout=drop, relu_dropout=drop, residual_dropout=drop, attention_dropout=drop))
    Pn = perturb(rt.init_params(hp, "transformer", seed=3), rng)
    src = rng.integers(3, hp.src_vocab.size(), (B, Ls)); src[:, -1] = 2
    tgt = rng.integers(3, hp.tgt_vocab.size(), (B, Lt)); tgt[:, -1] = 2
    src[1, Ls // 2:] = 0; src[1, Ls // 2 - 1] = 2          # a padded sentence: key masks inside the program
    a = _run(hp, Pn, src, tgt, False)
    b = _run(hp, Pn, src, tgt, True)
    assert b[5], "no layer program was built"
    strays, aborted = b[4]
    print("%s drop=%.1f: loss %.6f / %.6f, workgroups off their XCD: %d" % (cfg, drop, a[0], b[0], strays))
    assert not aborted
    assert torch.equal(a[2], b[2]), "first sub-layer differs"
    assert torch.equal(a[1], b[1]), "encoder output differs"
    assert a[0] == b[0]
    assert torch.equal(a[3], b[3]), "gradients differ"


def test_program_falls_back_for_uncovered_shapes():
    # 5 sentences of 9 rows: a group's rows are not whole 64-row tiles -> ordinary launches, same results as ever
    rng = np.random.default_rng(6)
    hp = make_hp("transformer")
    Pn = perturb(rt.init_params(hp, "transformer", seed=3), rng)
    src = rng.integers(3, 100, (5, 9)); src[:, -1] = 2
    tgt = rng.integers(3, 100, (5, 7)); tgt[:, -1] = 2
    a = _run(hp, Pn, src, tgt, False)
    b = _run(hp, Pn, src, tgt, True)
    assert not b[5]                       # nothing cached: the recording was rejected
    assert a[0] == b[0] and torch.equal(a[1], b[1])
