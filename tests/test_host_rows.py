# coding: utf-8
"""Host rows of SURVEY.md §8(f): learning-rate schedules, BLEU/OTEM/UTEM, the batch queue, the
checkpoint bundle + rotation, the evaluation helpers -- against golden vectors produced by the
reference's own TF-free modules (tests/golden/make_host_golden.py) where those exist."""
import json
import os

import numpy as np
import pytest

from tests.common import make_hp
from zero_amd import lrs
from zero_amd.utils import bundle, metric, queuer

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_host.json")))


def _make_lr(name, args):
    if name == "noam":
        return lrs.NoamDecayLr(*args)
    if name == "gnmt+":
        return lrs.GNMTPDecayLr(*args)
    if name.startswith("cosine"):
        return lrs.CosineDecayLr(args[0], args[1], args[2], args[3], args[4], t_mult=args[5], update_period=args[6])
    if name == "epoch":
        return lrs.EpochDecayLr(*args)
    if name == "score":
        return lrs.ScoreDecayLr(args[0], args[1], args[2], decay=args[3], patience=args[4])
    return lrs.VanillaLR(*args)


@pytest.mark.parametrize("name", sorted(GOLD["lrs"]))
def test_lr_schedules_match_reference_traces(name):
    case = GOLD["lrs"][name]
    lr = _make_lr(name, case["args"])
    for event, arg, want in case["trace"]:
        getattr(lr, event)(arg)
        assert lr.get_lr() == pytest.approx(want, rel=1e-12, abs=0.0), (name, event, arg)


def test_get_lr_factory_dispatch():
    hp = make_hp("transformer")
    kinds = {"noam": lrs.NoamDecayLr, "gnmt+": lrs.GNMTPDecayLr, "epoch": lrs.EpochDecayLr,
             "score": lrs.ScoreDecayLr, "vanilla": lrs.Lr, "cosine": lrs.CosineDecayLr}
    for name, cls in kinds.items():
        hp.lrate_strategy = name
        assert type(lrs.get_lr(hp)) is cls
    hp.lrate_strategy = "nope"
    with pytest.raises(NotImplementedError):
        lrs.get_lr(hp)
    # history replay accepts the recorder's (step, score) pairs (main.py:397)
    sc = lrs.ScoreDecayLr(1.0, 1e-3, 2.0, history_scores=[(10, 5.0), (20, 4.0), (30, 3.0)], decay=0.5, patience=1)
    assert sc.get_lr() == 0.25


@pytest.mark.parametrize("idx", range(len(GOLD["metric"])))
def test_bleu_otem_utem_match_reference(idx):
    case = GOLD["metric"][idx]
    for bp in ("closest", "shortest"):
        for smooth in (False, True):
            want = case["%s_%s" % (bp, "smooth" if smooth else "plain")]
            assert metric.bleu(case["cand"], case["refs"], bp=bp, smooth=smooth) == pytest.approx(want["bleu"], rel=1e-12, abs=1e-300)
            assert metric.otem(case["cand"], case["refs"], bp=bp, smooth=smooth) == pytest.approx(want["otem"], rel=1e-12, abs=1e-300)
            assert metric.utem(case["cand"], case["refs"], bp=bp, smooth=smooth) == pytest.approx(want["utem"], rel=1e-12, abs=1e-300)


def test_bleu_known_answers():
    s = ["a", "b", "c", "d", "e"]
    assert metric.bleu([s], [[s]]) == GOLD["metric_identity_bleu"] == 1.0
    assert metric.bleu([], []) == 0.0
    # one wrong token out of 5: p1 = 4/5, p2 = 2/4, p3 = 1/3 ... computed by hand
    got = metric.bleu([["a", "b", "x", "d", "e"]], [[s]], n=2)
    assert got == pytest.approx((0.8 * 0.5) ** 0.5, rel=1e-12)


@pytest.mark.parametrize("workers", [0, 1, 3])
def test_enqueuer_delivers_what_the_reference_delivers(workers):
    def reader():
        for i in range(23):
            yield [i, i * i]
    got = list(queuer.EnQueuer(reader(), lambda c: [c[0], c[1] + 1], worker_processes_num=workers,
                               input_queue_size=4, output_queue_size=4))
    want = GOLD["queuer"][str(workers)]
    assert (got if workers < 2 else sorted(got)) == want


def test_enqueuer_errors():
    with pytest.raises(ValueError):
        queuer.EnQueuer(iter([]), lambda x: x, worker_processes_num=-1)

    def bad():
        yield 1
        raise RuntimeError("reader broke")
    with pytest.raises(RuntimeError):
        list(queuer.EnQueuer(bad(), lambda x: x, worker_processes_num=1))
    assert list(queuer.EnQueuer(iter([]), lambda x: x, worker_processes_num=2)) == []


def test_device_feeder_on_cpu_keeps_order_and_dtypes():
    batches = [{"src": np.full((2, 3), i, np.int64), "tgt": np.full((2, 4), i + 1, np.int64), "index": [i]}
               for i in range(5)]
    seen = []
    for raw, dev in queuer.DeviceFeeder(iter(batches), "cpu"):
        assert dev["source"].dtype.is_floating_point is False and str(dev["source"].dtype) == "torch.int32"
        assert dev["source"].shape == (2, 3) and dev["target"].shape == (2, 4)
        seen.append((raw["index"][0], int(dev["source"][0, 0]), int(dev["target"][0, 0])))
    assert seen == [(i, i, i + 1) for i in range(5)]


# ---- checkpoint bundle -------------------------------------------------------------------
def test_crc32c_known_answers():
    assert bundle._crc32c_py(b"123456789") == 0xe3069283          # the standard CRC-32C check value
    assert bundle._crc32c_py(b"") == 0
    assert bundle._crc32c_py(bytes(32)) == 0x8a9136aa             # RFC 3720 B.4: 32 bytes of zeros
    assert bundle._crc32c_py(bytes([0xff] * 32)) == 0x62a8ab43    # RFC 3720 B.4: 32 bytes of ones
    assert bundle.unmask_crc(bundle.mask_crc(0xdeadbeef)) == 0xdeadbeef
    big = np.frombuffer(np.random.default_rng(0).bytes(70001), dtype=np.uint8)
    assert bundle.crc32c(big) == bundle._crc32c_py(big.tobytes())  # C-ABI zk_crc32c when built


def test_bundle_round_trip_and_structure(tmp_path):
    rng = np.random.default_rng(3)
    tensors = {"transformer/encoder/layer_0/x/W_0_0": rng.standard_normal((7, 5)).astype(np.float32),
               "transformer/bias": rng.standard_normal(5).astype(np.float32),
               "global_step": np.array(1234, dtype=np.int64),
               "beta1_power": np.array(0.5, dtype=np.float32),
               "half": rng.standard_normal(6).astype(np.float16),
               "empty": np.zeros((0, 4), np.float32)}
    for i in range(400):                                  # several index blocks
        tensors["transformer/decoder/layer_%03d/some/long/variable/name/W_0_0" % i] = rng.standard_normal(3).astype(np.float32)
    prefix = str(tmp_path / "model-1234")
    bundle.save_checkpoint(prefix, tensors)
    raw = open(prefix + ".index", "rb").read()
    assert raw[-8:] == bytes.fromhex("57fb808b247547db")     # table magic, little endian
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(t.nbytes for t in tensors.values())
    got = bundle.load_checkpoint(prefix, verify=True)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    listed = bundle.list_variables(prefix)
    assert [n for n, _, _ in listed] == sorted(tensors, key=lambda s: s.encode())
    assert dict((n, s) for n, s, _ in listed)["transformer/encoder/layer_0/x/W_0_0"] == (7, 5)
    sub = bundle.load_checkpoint(prefix, names={"global_step"})
    assert list(sub) == ["global_step"] and int(sub["global_step"]) == 1234
    # corruption is detected: flip one byte of a tensor, then one byte of the index
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(3); b = f.read(1); f.seek(3); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError):
        bundle.load_checkpoint(prefix, verify=True)
    with open(prefix + ".index", "r+b") as f:
        f.seek(10); b = f.read(1); f.seek(10); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError):
        bundle.load_checkpoint(prefix)


def test_bundle_reads_a_hand_assembled_index(tmp_path):
    """An index assembled byte by byte from the published format (independent of the writer):
    header key, one fp32 [2,2] tensor entry with explicit proto bytes."""
    import struct
    data = np.arange(4, dtype="<f4").tobytes()
    prefix = str(tmp_path / "m")
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    crc = bundle.mask_crc(bundle._crc32c_py(data))
    entry = bytes([0x08, 0x01,                     # dtype = DT_FLOAT
                   0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x02,   # shape { dim{size:2} dim{size:2} }
                   0x28, 0x10,                     # size = 16 (offset 0, shard 0 omitted)
                   0x35]) + struct.pack("<I", crc)
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])
    block = bytearray()
    block += bytes([0, 0, len(header)]) + header                       # key ""
    block += bytes([0, 3, len(entry)]) + b"v/w" + entry                # key "v/w"
    block += struct.pack("<III", 0, 0, 1)[4:]                          # restart[0] = 0, count = 1
    def framed(b):
        return bytes(b) + b"\x00" + struct.pack("<I", bundle.mask_crc(bundle._crc32c_py(bytes(b) + b"\x00")))
    meta = struct.pack("<II", 0, 1)
    idx_entry = bytearray(bytes([0, 3, 2]) + b"v/w" + bytes([0, len(block)]))
    index = idx_entry + struct.pack("<II", 0, 1)
    out = framed(block)
    meta_off = len(out); out += framed(meta)
    idx_off = len(out); out += framed(index)
    footer = bytes([meta_off, len(meta), idx_off, len(index)])
    footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    open(prefix + ".index", "wb").write(out + footer)
    got = bundle.load_checkpoint(prefix, verify=True)
    assert list(got) == ["v/w"] and got["v/w"].tolist() == [[0.0, 1.0], [2.0, 3.0]]


def test_saver_rotation_and_best(tmp_path):
    from zero_amd.utils.saver import Saver
    out = str(tmp_path / "run")
    os.makedirs(out)
    open(os.path.join(out, "param.json"), "w").write("{}")
    sv = Saver(checkpoints=2, output_dir=out, best_checkpoints=2)
    t = lambda k: {"s/w": np.full(3, k, np.float32), "global_step": np.array(k, np.int64)}
    sv.save(t(10), 10)
    sv.save(t(20), 20, metric_score=5.0)
    sv.save(t(30), 30, metric_score=7.0)
    sv.save(t(40), 40, metric_score=6.0)
    sv.save(t(50), 50, metric_score=1.0)
    lines = open(os.path.join(out, "checkpoint")).read().splitlines()
    assert lines[0] == 'model_checkpoint_path: "model-50"'
    assert lines[1:] == ['all_model_checkpoint_paths: "model-40"', 'all_model_checkpoint_paths: "model-50"']
    assert not os.path.exists(os.path.join(out, "model-10.index")) and os.path.exists(os.path.join(out, "model-50.meta"))
    best = os.path.join(out, "best")
    assert open(os.path.join(best, "metric.log")).read().splitlines() == ["Steps 20, Metric Score 5.0", "Steps 30, Metric Score 7.0"]
    assert open(os.path.join(best, "topk_checkpoint")).read().splitlines() == ["model-40\t6.0", "model-30\t7.0"]
    assert not os.path.exists(os.path.join(best, "model-20.index")) and os.path.exists(os.path.join(best, "model-30.index"))
    assert os.path.exists(os.path.join(best, "param.json"))
    assert open(os.path.join(best, "checkpoint")).readline().strip() == 'model_checkpoint_path: "model-30"'
    # a fresh Saver on the same directory resumes the state (saver.py:25-66)
    sv2 = Saver(checkpoints=2, output_dir=out, best_checkpoints=2)
    assert sv2.best_score == 7.0 and sv2.topk_scores == [("model-40", 6.0), ("model-30", 7.0)]
    got = sv2.restore()
    assert int(got["global_step"]) == 50 and got["s/w"].tolist() == [50.0] * 3
    assert Saver(output_dir=str(tmp_path / "none")).restore() is None


def test_store_checkpoint_names_round_trip(tmp_path):
    """Parameters + Adam slots under the reference's variable names; name-matching restore."""
    from zero_amd.variables import VariableStore
    from zero_amd.utils.saver import collect_tensors, assign_tensors
    hp = make_hp("transformer_aan", H=16, F=32, heads=2, layers=1, Vs=13, Vt=11)
    st = VariableStore(hp, "transformer_aan", "cpu")
    rng = np.random.default_rng(0)
    st.load({n: rng.standard_normal(st.lshape[n]).astype(np.float32) for n in st.names()})
    st.m.normal_(); st.v.uniform_()
    tensors = collect_tensors(st, "transformer", 77, hp)
    assert "transformer/decoder/layer_0/average_attention/z_project/W_0_0" in tensors
    assert "transformer/encoder/layer_0/self_attention/dot_attention/qkv_map/W_0_0/Adam_1" in tensors
    assert tensors["global_step"].dtype == np.int64 and tensors["beta2_power"] == np.float32(hp.beta2 ** 1)
    prefix = str(tmp_path / "model-77")
    bundle.save_checkpoint(prefix, tensors)
    st2 = VariableStore(hp, "transformer_aan", "cpu")
    loaded = bundle.load_checkpoint(prefix)
    del loaded["transformer/bias"]                   # a variable the checkpoint lacks stays as it is
    got, missing, step = assign_tensors(st2, "transformer", loaded)
    assert missing == ["bias"] and step == 77 and len(got) == len(st.names()) - 1
    for which in ("master", "m", "v"):
        a, b = st.export(which), st2.export(which)
        for n in st.names():
            if n != "bias":
                assert np.array_equal(a[n], b[n]), (which, n)


# ---- evaluation helpers ------------------------------------------------------------------
def test_decode_hypothesis_cuts_at_eos_or_pad():
    from zero_amd import evalu
    from zero_amd.vocab import Vocab
    v = Vocab()
    for w in ("a", "b", "c"):
        v.insert(w)
    hp = make_hp("transformer")
    hp.tgt_vocab = v
    a, b, c = v.get_id("a"), v.get_id("b"), v.get_id("c")
    seqs = np.array([[[a, b, v.eos(), c], [c, c, c, c]], [[b, v.pad(), a, a], [a, a, a, a]], [[c, c, c, c], [a, a, a, a]]])
    scores = np.array([[-1.0, -2.0], [-3.0, -4.0], [-0.5, -9.0]])
    hyp, marks = evalu.decode_hypothesis([seqs], [scores], hp)
    assert hyp == [["a", "b"], ["b"], ["c", "c", "c", "c"]] and marks == [-1.0, -3.0, -0.5]
    assert evalu.decode_hypothesis([seqs], [scores], hp, mask=[0.]) == ([], [])


def test_eval_metric_and_dump(tmp_path):
    from zero_amd import evalu
    ref = tmp_path / "dev.tgt"
    ref.write_text("a b c d e\nx y z\n")
    trans = [["x", "y", "z"], ["a", "b", "c", "d", "e"]]
    assert evalu.eval_metric(trans, str(ref), indices=[1, 0]) == 1.0
    assert evalu.eval_metric(trans, str(tmp_path / "missing")) == 0.0
    (tmp_path / "multi.ref0").write_text("a b c d e\nq q q\n")
    (tmp_path / "multi.ref1").write_text("e d c b a\nx y z\n")
    assert evalu.eval_metric(trans, str(tmp_path / "multi"), indices=[1, 0]) == 1.0
    out = tmp_path / "o" / "t.txt"
    evalu.dump_tanslation(trans, str(out), indices=[1, 0])
    assert out.read_text() == "a b c d e\nx y z\n"
    evalu.dump_tanslation([0.5, 1.5], str(out))
    assert out.read_text() == "0.5\n1.5\n"


def test_scoring_and_decoding_loops_with_stub_paths(tmp_path):
    """The loops' bookkeeping (length-sorted batches -> file order, ppl) with stub score / infer
    functions standing in for the GPU paths."""
    from zero_amd import evalu
    from zero_amd.data import Dataset
    from zero_amd.vocab import Vocab
    src = tmp_path / "s.txt"; tgt = tmp_path / "t.txt"
    src.write_text("a b c d\na\na b\n"); tgt.write_text("b\nb c d\nc c\n")
    v = Vocab()
    for w in "abcd":
        v.insert(w)
    hp = make_hp("transformer")
    hp.src_vocab = hp.tgt_vocab = v
    hp.eval_batch_size, hp.buffer_size, hp.process_num = 2, 10, 0
    ds = Dataset(str(src), str(tgt), v, v, 100, batch_or_token='batch')
    score = lambda feats, graph, params: (feats["target"] > 0).sum(1).astype(np.float32) * 0.5
    scores, ppl = evalu.scoring(None, ds, hp, score=score)
    assert scores == [1.0, 2.0, 1.5]          # file order; lengths include eos: 2, 4, 3
    assert ppl == pytest.approx(np.exp((1.0 * 2 + 2.0 * 4 + 1.5 * 3) / 9.0))

    def infer(feats, graph, params):           # "translate" = copy the source, beam of 1
        s = feats["source"]
        return s[:, None, :], -np.arange(len(s), dtype=np.float32)[:, None]
    ds2 = Dataset(str(src), str(src), v, v, 100, batch_or_token='batch')
    tr, sc, idx = evalu.decoding(None, ds2, hp, infer=infer)
    ordered = [t for _, t in sorted(zip(idx, tr))]
    assert ordered == [["a", "b", "c", "d"], ["a"], ["a", "b"]]


def test_recorder_round_trip(tmp_path):
    from zero_amd.utils.recorder import new_recorder, Recorder
    hp = make_hp("transformer")
    r = new_recorder(hp)
    r.step, r.valid_script_scores = 12, [(10, 3.5)]
    p = str(tmp_path / "record.json")
    r.save_to_json(p)
    r2 = Recorder(); r2.load_from_json(p)
    assert r2.step == 12 and r2.valid_script_scores == [[10, 3.5]] and r2.epoch == 1 and r2.lidx == -1


def test_checkpoint_averaging(tmp_path):
    """scripts/checkpoint_averaging.py: newest N of the run, float64 mean, global_step reset, json copied."""
    from zero_amd.scripts import checkpoint_averaging as ca
    from zero_amd.utils.saver import Saver
    run = tmp_path / "run"; run.mkdir()
    (run / "param.json").write_text('{"hidden_size": 8}')
    sv = Saver(checkpoints=5, output_dir=str(run))
    for step in (10, 20, 30, 40):
        sv.save({"s/w": np.full((2, 3), float(step), np.float32), "s/h": np.full(4, step, np.float16),
                 "global_step": np.array(step, np.int64)}, step)
    out = tmp_path / "avg"
    used = ca.average(str(run), 3, str(out))
    assert [os.path.basename(p) for p in used] == ["model-40", "model-30", "model-20"]
    got = bundle.load_checkpoint(str(out / "average-0"))
    assert got["s/w"].dtype == np.float32 and np.allclose(got["s/w"], 30.0) and got["s/h"].dtype == np.float16
    assert int(got["global_step"]) == 0 and (out / "param.json").exists()
    assert Saver(output_dir=str(out)).restore()["s/w"].shape == (2, 3)
    ca.main(["--path", str(run), "--checkpoints", "10", "--output", str(tmp_path / "avg2")])
    assert np.allclose(bundle.load_checkpoint(str(tmp_path / "avg2" / "average-0"))["s/w"], 25.0)
    with pytest.raises(ValueError):
        ca.get_checkpoints(str(tmp_path / "nowhere"))


def test_vocab_and_vocabulary_preparation_match_the_reference(tmp_path):
    """vocab.py: the preparation tool (count, sort by falling frequency with first-seen ties, cut to --size), loading,
    to_id / to_tokens incl. unknown tokens and out-of-range ids, reserved ids -- against outputs of the reference's
    own module on seeded corpora (tests/golden/reference_vocab.json, made by tests/golden/make_vocab_golden.py)."""
    import json
    from zero_amd.vocab import Vocab, main as vocab_main
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vocab.json")))["cases"]
    assert len(cases) >= 6
    for i, c in enumerate(cases):
        src, out = tmp_path / ("corpus%d.txt" % i), tmp_path / ("vocab%d.txt" % i)
        src.write_text("\n".join(c["corpus"]) + ("\n" if c["corpus"] else ""))
        argv = [str(src), str(out)] + ([] if c["size"] >= 1e6 else ["--size", str(int(c["size"]))])
        vocab_main(argv)
        assert out.read_text() == c["vocab_file"], i
        v = Vocab(str(out))
        assert v.size() == c["loaded_size"]
        assert v.to_id(c["probe"]) == c["to_id"] and v.to_id(c["probe"], append_eos=False) == c["to_id_no_eos"]
        assert v.to_tokens(c["ids"]) == c["to_tokens"]
        assert (v.eos(), v.pad()) == (c["eos"], c["pad"]) == (2, 0)


# ---- round 5: the hypothesis cut (evalu.py:14-46) and closing_dropout (util.py:106-114) against the reference's own
# function bodies (tests/golden/make_evalu_golden.py lifts them out of the syntax trees; the modules import TensorFlow)
def test_hypothesis_cut_and_closing_dropout_equal_the_reference_functions(tmp_path):
    import json as _json
    import os as _os
    from zero_amd import evalu
    from zero_amd.models._factory import closing_dropout
    from zero_amd.search import decode_hypothesis as cut_best
    from zero_amd.utils.hparams import HParams
    from zero_amd.vocab import Vocab
    fx = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "reference_evalu.json")))
    vp = tmp_path / "v.txt"
    vp.write_text("\n".join(fx["vocab_lines"]) + "\n")
    v = Vocab(str(vp))

    class P(object):
        tgt_vocab = v
    assert len(fx["hypothesis"]) >= 12
    for c in fx["hypothesis"]:
        hyp, marks = evalu.decode_hypothesis(c["seqs"], c["scores"], P(), mask=c["mask"])
        assert hyp == c["hypoes"] and marks == c["marks"], c
        # the id-level cut the search tests use (beam 0, stop at the first eos or pad) is the same rule
        for t, tower in enumerate(c["seqs"]):
            ids = cut_best(np.asarray(tower), P())
            assert [v.to_tokens(i) for i in ids] == [evalu.decode_target_token(s[0], v) for s in tower]
    for c in fx["closing_dropout"]:
        hp = HParams(**c["before"])
        assert closing_dropout(hp).values() == c["after"], c
