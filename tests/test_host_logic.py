"""Host-side contract: HParams / config / registry / vocab / variable layout / LR schedule."""
import copy
import json
import os

import numpy as np
import pytest

from zero_amd.config import default_params, transformer_base_params, SyntheticVocab
from zero_amd.utils.hparams import HParams
from zero_amd.models import model as registry, load_all
from zero_amd import lrs
from zero_amd.vocab import Vocab
from zero_amd.variables import VariableStore, variable_specs, initial_values, ALIGN
from tests.common import make_hp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_defaults_follow_run_py():
    p = default_params()
    v = p.values()
    assert len(v) == 99 + 0 or len(v) >= 95
    assert p.hidden_size == 1000 and p.embed_size == 620 and p.filter_size == 2048
    assert p.beam_size == 4 and p.decode_alpha == 0.6 and p.decode_length == 50
    assert p.search_mode == "cache" and p.aan_mask is True and p.use_ffn is False
    assert p.clip_grad_norm == 5.0 and p.epsilon == 1e-9 and p.beta2 == 0.999
    assert p.dtype_epsilon == 1e-8 and p.dtype_inf == 1e8 and p.gpus == [0] and p.strategies == ["aan"]


def test_parse_override_json_roundtrip():
    p = default_params()
    p.parse("hidden_size=512,gpus=[0,1,2],shared_source_target_embedding=true,lrate=0.5,model_name=transformer")
    assert p.hidden_size == 512 and p.gpus == [0, 1, 2] and p.shared_source_target_embedding is True
    assert p.lrate == 0.5 and p.model_name == "transformer"
    p.override_from_dict(dict(num_heads=16, clip_grad_norm=0.0))
    assert p.num_heads == 16 and p.clip_grad_norm == 0.0
    js = p.to_json()
    q = default_params()
    q.parse_json(js)
    assert q.values() == p.values()
    with pytest.raises(ValueError):
        p.parse("no_such_key=1")
    with pytest.raises(ValueError):
        p.parse("hidden_size=1.5")
    p.add_hparam("recorder", {"step": 3})
    assert p.recorder["step"] == 3
    c = copy.copy(p)
    c.hidden_size = 7
    assert p.hidden_size == 512 and c.num_heads == 16
    p.src_vocab = SyntheticVocab(10)        # plain attribute, not an hparam (run.py:386)
    assert "src_vocab" not in p.values() and "src_vocab" not in json.loads(p.to_json())


def test_registry_contract():
    load_all()
    for name in ("transformer", "transformer_aan", "transformer_rpr", "transformer_fuse", "TRANSFORMER"):
        m = registry.get_model(name)
        assert callable(m.train_fn) and callable(m.score_fn) and callable(m.infer_fn)
    with pytest.raises(Exception) as e:
        registry.get_model("nope")
    assert "No supported model" in str(e.value)
    with pytest.raises(Exception) as e:
        registry.model_register("Transformer", None, None, None)
    assert "Conflict Model Name" in str(e.value)


def test_vocab_and_noam_against_reference_values():
    gold = json.load(open(os.path.join(GOLD, "reference_scalars.json")))
    v = Vocab()
    assert {"pad": v.pad(), "eos": v.eos(), "unk": v.get_id("<unk>"), "size": v.size()} == gold["vocab_ids"]
    v.insert("hello"); v.insert("world")
    assert v.to_id(["hello", "zzz", "world"]) == gold["vocab_to_id"]
    hp = make_hp("transformer", H=512, lrate=1.0, warmup_steps=4000)
    lr = lrs.get_lr(hp)
    for step, val in gold["noam_lr_init1_warm4000_h512"].items():
        lr.step(int(step))
        assert lr.get_lr() == pytest.approx(val, rel=1e-12)
    hp2 = make_hp("transformer", H=1024, lrate=2.0, warmup_steps=400, min_lrate=1e-5, max_lrate=1e-3)
    lr = lrs.get_lr(hp2)
    for step, val in gold["noam_lr_init2_clamped_warm400_h1024"].items():
        lr.step(int(step))
        assert lr.get_lr() == pytest.approx(val, rel=1e-12)


@pytest.mark.parametrize("model", ["transformer", "transformer_aan", "transformer_rpr", "transformer_fuse"])
def test_variable_names_match_the_oracles(model):
    from oracle import ref_torch as rt
    hp = make_hp(model, Vs=13, Vt=11)
    ours = [(n, tuple(s)) for n, s, _, _ in variable_specs(hp, model)]
    theirs = [(n, tuple(s)) for n, s, _, _ in rt.variable_specs(hp, model)]
    assert ours == theirs
    assert "encoder/layer_0/self_attention/dot_attention/qkv_map/W_0_0" in dict(ours)
    assert ("bias", (hp.embed_size,)) in ours


def test_base_parameter_count_and_layout():
    hp = transformer_base_params()
    hp.src_vocab = SyntheticVocab(32000); hp.tgt_vocab = SyntheticVocab(32000)
    specs = variable_specs(hp, "transformer")
    n = sum(int(np.prod(s)) for _, s, _, _ in specs)
    assert abs(n - 76.9e6) < 0.1e6          # 44.14M body + 2 x 16.38M embeddings
    hp = make_hp("transformer", Vs=13, Vt=11)
    st = VariableStore(hp, "transformer", "cpu")
    assert st.pshape["tgt_embedding"] == (16, hp.embed_size) and st.lshape["tgt_embedding"][0] == 11
    for name, off in st.offsets.items():
        assert off % ALIGN == 0
    vals = initial_values(hp, "transformer", 3)
    st.load(vals)
    back = st.export("master")
    for k in vals:
        assert np.array_equal(back[k], vals[k])
    assert float(st.w("tgt_embedding")[11:].abs().max()) == 0.0   # physical pad rows stay zero
    emb = vals["src_embedding"]
    assert abs(emb.std() - hp.hidden_size ** -0.5) < 0.02
    w = vals["encoder/layer_0/feed_forward/ffn_layer/enlarge/W_0_0"]
    lim = np.sqrt(3.0 / ((w.shape[0] + w.shape[1]) / 2.0))
    assert np.abs(w).max() <= lim + 1e-6 and np.abs(w).max() > 0.9 * lim


def test_trim_columns_is_remove_invalid_seq():
    from zero_amd.models._core import trim_columns
    assert trim_columns(np.zeros((2, 3), dtype=np.int64)).shape == (2, 1)
    assert trim_columns(np.array([[4, 2, 0, 0], [5, 6, 2, 0]])).shape == (2, 3)


def test_beam_host_step_in_c_matches_the_numpy_bookkeeping():
    """zk_beam_host_step / zk_beam_host_should_stop (pure host C in libzero_hip.so) against the numpy
    statements of search.py:85-113,168-228 used by the eager / dev search: random survivors incl. exact
    ties, EOS symbols and length caps, several steps in a row."""
    import ctypes
    from zero_amd import hip
    lib = hip.lib()
    f32 = np.float32
    F32_MIN = np.finfo(np.float32).min
    rng = np.random.default_rng(5)
    B, K, V, eos, pad, alpha = 5, 3, 50, 2, 0, 0.6
    Tcap = 12
    mtl = np.array([3, 9, 9, 2, 9], dtype=f32); mtl_i = mtl.astype(np.int32)

    def top_k(x, k):
        idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
        return np.take_along_axis(x, idx, axis=-1), idx
    # numpy state
    seq = np.full((B, K, 1), pad, dtype=np.int64); fin_seq = np.zeros_like(seq)
    log_probs = np.tile(np.array([[0.] + [F32_MIN] * (K - 1)], dtype=f32), (B, 1)); scores = np.zeros_like(log_probs)
    fin_scores = np.full((B, K), F32_MIN, dtype=f32); fin_flags = np.zeros((B, K), dtype=bool)
    # C state
    c_seq = np.full((B, K, Tcap), pad, dtype=np.int32); c_fin = np.zeros((B, K, Tcap), dtype=np.int32)
    c_lp = log_probs.copy(); c_sc = scores.copy(); c_fs = fin_scores.copy(); c_ff = np.zeros((B, K), dtype=np.uint8)
    c_idx = np.zeros(B * K, np.int32); c_tok = np.zeros(B * K, np.int32)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    for time in range(8):
        max_lp = np.power((f32(5.) + mtl) / f32(6.), f32(alpha)).astype(f32)
        worst = (fin_scores * fin_flags.astype(f32)).min(axis=1) + (f32(1.) - fin_flags.any(axis=1).astype(f32)) * F32_MIN
        want_stop = bool((worst > log_probs[:, 0] / max_lp).all()) or not bool((time < mtl_i).any())
        got_stop = lib.raw("zk_beam_host_should_stop")(B, K, P(c_lp), P(c_fs), P(c_ff), P(mtl), P(mtl_i), time, alpha)
        assert bool(got_stop) == want_stop
        if want_stop:
            break
        penalty = f32(np.power(f32((f32(5.) + f32(time + 1)) / f32(6.)), f32(alpha)))
        ts = np.sort(rng.standard_normal((B, 2 * K)).astype(f32) * 3 - time, axis=1)[:, ::-1].copy()
        ts[0, 1] = ts[0, 0]                                              # an exact tie
        ti = rng.integers(0, K * V, (B, 2 * K)).astype(np.int32)
        ti[1, 0] = 1 * V + eos; ti[2, 3] = 0 * V + eos                   # some EOS symbols
        # ---- numpy (search.py:168-228 as in zero_amd/search.py)
        beam_idx = ti.astype(np.int64) // V; sym = ti.astype(np.int64) % V
        bpos = np.arange(B)[:, None]
        curr_seq = np.concatenate([seq[bpos, beam_idx], sym[:, :, None]], axis=2)
        curr_fin = (sym == eos) | (time >= mtl_i)[:, None]
        with np.errstate(over="ignore"):
            alive_scores, alive_idx = top_k(ts + curr_fin.astype(f32) * F32_MIN, K)
            alive_lp = (alive_scores * penalty).astype(f32)
            cfs = ts + (f32(1.) - curr_fin.astype(f32)) * F32_MIN
        all_flags = np.concatenate([fin_flags, curr_fin], axis=1); all_scores = np.concatenate([fin_scores, cfs], axis=1)
        fin_scores, fin_idx = top_k(all_scores, K)
        fin_flags = all_flags[bpos, fin_idx]
        all_seq = np.concatenate([np.concatenate([fin_seq, np.full((B, K, 1), pad, dtype=seq.dtype)], axis=2), curr_seq], axis=1)
        fin_seq = all_seq[bpos, fin_idx]
        flat = (np.arange(B)[:, None] * K + beam_idx[bpos, alive_idx]).reshape(-1)
        seq, log_probs, scores = curr_seq[bpos, alive_idx], alive_lp, alive_scores
        # ---- C
        rc = lib.raw("zk_beam_host_step")(B, K, V, Tcap, time, P(ts), P(ti), P(c_seq), P(c_fin), P(c_lp), P(c_sc), P(c_fs),
                                          P(c_ff), P(mtl_i), eos, pad, float(penalty), P(c_idx), P(c_tok))
        assert rc == 0
        n = time + 2
        assert np.array_equal(c_seq[:, :, :n], seq) and np.array_equal(c_fin[:, :, :n], fin_seq)
        assert np.array_equal(c_lp.view(np.int32), log_probs.view(np.int32))          # bit-exact fp32
        assert np.array_equal(c_sc.view(np.int32), scores.view(np.int32))
        assert np.array_equal(c_fs.view(np.int32), fin_scores.view(np.int32))
        assert np.array_equal(c_ff.astype(bool), fin_flags)
        assert np.array_equal(c_idx, flat) and np.array_equal(c_tok, seq[:, :, -1].reshape(-1))
    assert time >= 3


def test_config_file_values_may_be_arithmetic_but_never_code(tmp_path):
    """run.py:367-376 eval()s the --config file; here: literals, arithmetic over them and names of earlier keywords,
    nothing executable."""
    from zero_amd import run
    cfg = tmp_path / "cfg.py"
    cfg.write_text("dict(hidden_size=256, filter_size=hidden_size*4, lrate=1e-3*4, token_size=2**12, safe_nan=True and False,\n"
                   "     strategies=['aan'], max_len=-(-100), warmup_steps=8000//2)\n")
    hp = run.build_params("", str(cfg))
    assert (hp.hidden_size, hp.filter_size, hp.token_size, hp.max_len, hp.warmup_steps) == (256, 1024, 4096, 100, 4000)
    assert abs(hp.lrate - 4e-3) < 1e-12 and hp.strategies == ["aan"]
    for bad in ("dict(a=__import__('os').system('true'))", "dict(a=open('/etc/passwd'))", "dict(a=(1).__class__)",
                "dict(a=b)", "dict(a=[x for x in (1,)])", "dict(a=9**9**9)"):
        cfg.write_text(bad)
        try:
            run.build_params("", str(cfg))
        except (ValueError, SyntaxError):
            continue
        raise AssertionError("accepted: " + bad)


def test_decode_many_order_lanes_and_errors():
    """evalu.decode_many (host logic only): results in input order whatever lane finished first, every lane bound to its
    own execution lane index, each_lane warm-up runs every item on every lane, a worker's exception reaches the caller,
    and one stream is the plain sequential loop."""
    import threading
    import time
    from zero_amd.evalu import decode_many
    from zero_amd.models._factory import current_lane
    seen = {}
    lock = threading.Lock()

    def work(x):
        time.sleep(0.001 * (7 - x % 7))             # later items finish earlier
        with lock:
            seen.setdefault(current_lane(), []).append(x)
        return x * x
    items = list(range(23))
    assert decode_many(iter(items), work, streams=1) == [x * x for x in items] and set(seen) == {0}
    seen.clear()
    assert decode_many(iter(items), work, streams=3) == [x * x for x in items]
    assert set(seen) <= {0, 1, 2} and sorted(sum(seen.values(), [])) == items and len(seen) >= 2
    assert current_lane() == 0                      # the caller's lane is untouched
    seen.clear()
    assert decode_many(items[:4], work, streams=3, each_lane=True) == [0, 1, 4, 9]
    assert {k: sorted(v) for k, v in seen.items()} == {0: [0, 1, 2, 3], 1: [0, 1, 2, 3], 2: [0, 1, 2, 3]}

    def boom(x):
        if x == 5:
            raise KeyError("lane failure")
        return x
    try:
        decode_many(items, boom, streams=2)
    except KeyError:
        pass
    else:
        raise AssertionError("the worker's exception was swallowed")


def test_pinned_ring_cpu_path_and_bench_rotation():
    """utils/queuer.PinnedRing off the GPU is a plain copy (int64 ids -> int32); bench.py's rotation of synthetic batches:
    eight distinct batches per rank, batch 0 = the batch rounds 1-3 replayed, ranks draw different ones."""
    import numpy as np
    import torch
    from zero_amd.utils.queuer import PinnedRing
    ring = PinnedRing("cpu")
    dst = torch.zeros(3, 4, dtype=torch.int32)
    for i in range(20):                         # more puts than slots
        ring.put(dst, np.arange(12, dtype=np.int64).reshape(3, 4) + i)
        assert dst[2, 3].item() == 11 + i
    import bench
    b = [bench.synthetic_batch(0, None, i) for i in range(bench.ROTATION)]
    assert len({x[0].tobytes() for x in b}) == bench.ROTATION and all(x[0].shape == (64, 64) for x in b)
    rng = np.random.default_rng(1234)
    assert np.array_equal(b[0][0][:, :-1], rng.integers(3, 32000, size=(64, 64), dtype=np.int64)[:, :-1])
    assert not np.array_equal(bench.synthetic_batch(1, None, 0)[0], b[0][0])
    assert all((x[0][:, -1] == 2).all() and (x[1][:, -1] == 2).all() for x in b)


def test_headline_rule_of_the_multi_gpu_legs():
    """bench.choose_headline: the fastest reference-exact (fp32-bucket) leg unless a bf16-bucket leg
    wins by MORE than 3 %; skipped legs and a run whose optional legs never finished (watchdog) do not change the rule."""
    import bench
    L = lambda name, ms, exact: {"leg": name, "ms_per_step": ms, "reference_exact": exact}
    legs = [L("fp32/torch/dense", 5.00, True), L("fp32/torch/rows", 4.90, True), L("bf16/torch/rows", 4.80, False),
            {"leg": "fp32/zk_comm/rows", "skipped": "the direct communicator did not come up on every rank"}]
    assert bench.choose_headline(legs)["leg"] == "fp32/torch/rows"            # bf16 wins by 2 %: not enough
    legs[2]["ms_per_step"] = 4.70
    assert bench.choose_headline(legs)["leg"] == "bf16/torch/rows"            # 4.1 %: bf16 is the headline
    assert bench.choose_headline(legs[:1])["leg"] == "fp32/torch/dense"       # only the first leg finished
    assert bench.choose_headline([legs[3]]) is None and bench.choose_headline([]) is None
    assert bench.choose_headline([legs[2]]) is None                           # no reference-exact leg: no headline


def test_counter_files_of_an_earlier_round_or_a_dirty_tree_are_flagged_stale(monkeypatch):
    """bench.counter_status: PMC summaries can only be cited by a line; a citation of another round's
    file, of a file without a commit or of one measured on a dirty tree says so."""
    import bench
    monkeypatch.setattr(bench, "current_round", lambda: 5)
    assert bench.counter_status("r05_pmc_traffic.json", "abc123def456") == {"stale": False, "why": None}
    assert bench.counter_status("r04_pmc_traffic.json", "abc123def456")["stale"] is True
    assert bench.counter_status("r05_pmc_mfma.json", "abc123def456+dirty")["stale"] is True
    assert bench.counter_status("r05_pmc_mfma.json", None)["stale"] is True
    assert bench.counter_status("pmc.json", "abc")["stale"] is True and bench.counter_status(None, None) is None
    monkeypatch.undo()
    assert bench.current_round() >= 1
