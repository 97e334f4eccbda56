"""GPU tests of the callers either side of the step: EMA shadows, the train /
evaluate / score loops of zero_amd/main.py over a small on-disk bitext with the reference's
checkpoint layout, and the pinned double-buffered device feed."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.common import make_hp, make_batch  # noqa: E402
from zero_amd.models._factory import reset_cores  # noqa: E402
from zero_amd.models import load_all  # noqa: E402
from zero_amd.variables import reset_stores  # noqa: E402

load_all()


def test_ema_shadows_follow_tf_exponential_moving_average():
    from zero_amd.main import Trainer
    reset_cores(); reset_stores()
    hp = make_hp("transformer", ema_decay=0.99, lrate=0.05, warmup_steps=2)
    rng = np.random.default_rng(0)
    src, tgt = make_batch(rng, 4, 7, 8, hp.src_vocab.size(), hp.tgt_vocab.size())
    tr = Trainer(hp)
    name = "encoder/layer_0/feed_forward/ffn_layer/enlarge/W_0_0"
    ema = tr.store.export("master")[name].astype(np.float64)
    for step in range(1, 5):
        tr.micro_step({"source": src, "target": tgt})
        torch.cuda.synchronize()
        w = tr.store.export("master")[name].astype(np.float64)
        d = min(0.99, (1.0 + step) / (10.0 + step))          # tf.train.ExponentialMovingAverage(num_updates=global_step)
        ema = ema - (1.0 - d) * (ema - w)
        got = tr.store._view(tr.train_op.ema, name)
        ls = tr.store.lshape[name]
        assert np.abs(got[:ls[0], :ls[1]].cpu().numpy() - ema).max() < 1e-6
    raw = tr.store.export("master")[name]
    tr.train_op.ema_backup(); tr.train_op.ema_assign()
    assert np.abs(tr.store.export("master")[name] - ema).max() < 1e-6
    tr.train_op.ema_restore()
    assert np.array_equal(tr.store.export("master")[name], raw)


def test_device_feeder_pinned_double_buffer():
    from zero_amd.utils import queuer
    rng = np.random.default_rng(1)
    batches = [{"src": rng.integers(0, 100, (3 + i % 4, 5 + i)), "tgt": rng.integers(0, 100, (3 + i % 4, 4 + i)), "index": [i]}
               for i in range(9)]
    n = 0
    for raw, dev in queuer.DeviceFeeder(iter(batches), "cuda:0"):
        assert dev["source"].is_cuda and dev["source"].dtype == torch.int32
        assert np.array_equal(dev["source"].cpu().numpy(), raw["src"]) and np.array_equal(dev["target"].cpu().numpy(), raw["tgt"])
        n += 1
    assert n == 9


def _write_bitext(tmp, n=48, seed=3):
    """A copy task over a 12-word vocabulary: learnable in a few hundred tiny steps."""
    rng = np.random.default_rng(seed)
    words = ["w%d" % i for i in range(12)]
    (tmp / "vocab.txt").write_text("\n".join(words) + "\n")
    lines = [" ".join(rng.choice(words, size=int(rng.integers(2, 7)))) for _ in range(n)]
    for name in ("train.src", "train.tgt", "dev.src", "dev.tgt"):
        (tmp / name).write_text("\n".join(lines if name.startswith("train") else lines[:8]) + "\n")
    return lines


def test_train_eval_score_loops_on_disk(tmp_path):
    from zero_amd import main as loops, run as cli
    from zero_amd.utils import bundle
    reset_cores(); reset_stores()
    lines = _write_bitext(tmp_path)
    out = tmp_path / "out"
    kv = dict(hidden_size=32, embed_size=32, filter_size=64, num_heads=2, num_encoder_layer=1, num_decoder_layer=1,
              dropout=0.0, relu_dropout=0.0, residual_dropout=0.0, attention_dropout=0.0, label_smooth=0.1,
              model_name="transformer", scope_name="transformer", batch_or_token="batch", batch_size=16,
              eval_batch_size=8, max_training_steps=60, epoches=1000, disp_freq=20, save_freq=25, eval_freq=30,
              lrate=0.3, lrate_strategy="noam", warmup_steps=20, beam_size=2, decode_length=4, process_num=1,
              buffer_size=100, shuffle_batch=False, ema_decay=-1.0, checkpoints=2, best_checkpoints=1,
              src_vocab_file=str(tmp_path / "vocab.txt"), tgt_vocab_file=str(tmp_path / "vocab.txt"),
              src_train_file=str(tmp_path / "train.src"), tgt_train_file=str(tmp_path / "train.tgt"),
              src_dev_file=str(tmp_path / "dev.src"), tgt_dev_file=str(tmp_path / "dev.tgt"),
              src_test_file=str(tmp_path / "dev.src"), tgt_test_file=str(tmp_path / "dev.tgt"),
              output_dir=str(out), test_output=str(out / "test.trans.txt"), gpus=[0], random_seed=7)
    params = cli.setup(cli.build_params(",".join("%s=%s" % (k, v) for k, v in kv.items() if not isinstance(v, (list,)))))
    cli.save_parameters(params, params.output_dir)
    cli.setup_recorder(params)
    best = loops.train(params)
    # checkpoints in the reference's layout, rotation to `checkpoints`, best/ + logs
    state = (out / "checkpoint").read_text().splitlines()
    assert state[0] == 'model_checkpoint_path: "model-60"' and len(state) == 3
    names = dict((n, s) for n, s, _ in bundle.list_variables(str(out / "model-60")))
    assert names["transformer/encoder/layer_0/self_attention/dot_attention/qkv_map/W_0_0"] == (32, 96)
    assert "transformer/tgt_embedding/Adam_1" in names and names["global_step"] == ()
    assert (out / "best" / "metric.log").exists() and (out / "best" / "topk_checkpoint").exists()
    rec = json.load(open(out / "record.json"))
    assert rec["valid_script_scores"] and rec["valid_script_scores"][0][0] == 30
    assert (out / "eval-30.trans.txt").exists() and (out / "eval-60.trans.txt").exists()
    assert best == max(v[1] for v in rec["valid_script_scores"])
    # test / score modes restore the latest checkpoint into a fresh replica
    reset_cores(); reset_stores()
    p2 = cli.setup(cli.build_params("output_dir=%s" % out))
    bleu = loops.evaluate(p2)
    trans = (out / "test.trans.txt").read_text().splitlines()
    assert len(trans) == 8 and 0.0 <= bleu <= 1.0
    reset_cores(); reset_stores()
    p3 = cli.setup(cli.build_params("output_dir=%s,test_output=%s" % (out, out / "scores.txt")))
    mean_score = loops.scorer(p3)
    scores = [float(x) for x in (out / "scores.txt").read_text().split()]
    assert len(scores) == 8 and abs(np.mean(scores) - mean_score) < 1e-5
    # the restored replica scores exactly like the one that trained (same weights, same kernels)
    reset_cores(); reset_stores()
    p4 = cli.setup(cli.build_params("output_dir=%s,test_output=%s" % (out, out / "scores2.txt")))
    assert loops.scorer(p4) == mean_score
    # ... and a 60-step model has learnt something about copying: loss well under the uniform level
    assert mean_score < 0.95 * np.log(params.tgt_vocab.size())


def test_cli_train_test_score(tmp_path):
    """The reference's command line (run.py:241-246, 367-413) end to end in a fresh interpreter:
    --mode train on files, then --mode test and --mode score from the written param.json."""
    import subprocess
    import sys
    _write_bitext(tmp_path, n=32)
    out = tmp_path / "cli"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    cfg = tmp_path / "cfg.py"
    cfg.write_text("dict(hidden_size=32, embed_size=32, filter_size=64, num_heads=2, num_encoder_layer=1,\n"
                   "     num_decoder_layer=1, model_name='transformer_aan', batch_or_token='batch', batch_size=16,\n"
                   "     eval_batch_size=8, max_training_steps=12, epoches=100, disp_freq=6, save_freq=6, eval_freq=12,\n"
                   "     warmup_steps=10, lrate=0.3, lrate_strategy='noam', beam_size=2, decode_length=4, process_num=1,\n"
                   "     shuffle_batch=False, dropout=0.0, relu_dropout=0.0, residual_dropout=0.0, attention_dropout=0.0)\n")
    files = ",".join("%s=%s" % (k, tmp_path / v) for k, v in dict(
        src_vocab_file="vocab.txt", tgt_vocab_file="vocab.txt", src_train_file="train.src", tgt_train_file="train.tgt",
        src_dev_file="dev.src", tgt_dev_file="dev.tgt", src_test_file="dev.src", tgt_test_file="dev.tgt").items())
    base = [sys.executable, "-m", "zero_amd.run", "--config", str(cfg)]
    r = subprocess.run(base + ["--mode", "train", "--parameters", files + ",output_dir=%s" % out],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "best_score" in r.stdout
    saved = json.loads((out / "param.json").read_text())
    assert saved["model_name"] == "transformer_aan" and saved["hidden_size"] == 32
    assert (out / "checkpoint").exists() and (out / "record.json").exists() and (out / "eval-12.trans.txt").exists()
    r = subprocess.run(base + ["--mode", "test", "--parameters", "output_dir=%s,test_output=%s" % (out, out / "t.txt")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "bleu" in r.stdout and len((out / "t.txt").read_text().splitlines()) == 8
    # the same test set in the fp32 decode mode (round 5: decode_dtype=float32 through the CLI, four batches on lanes)
    r = subprocess.run(base + ["--mode", "test", "--parameters",
                               "output_dir=%s,test_output=%s,decode_dtype=float32" % (out, out / "t32.txt")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    t16, t32 = (out / "t.txt").read_text().splitlines(), (out / "t32.txt").read_text().splitlines()
    assert "bleu" in r.stdout and len(t32) == 8
    assert sum(a == b for a, b in zip(t16, t32)) >= 4          # (a barely trained 32-wide model: most lines agree)
    r = subprocess.run(base + ["--mode", "score", "--parameters", "output_dir=%s,test_output=%s" % (out, out / "s.txt")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len((out / "s.txt").read_text().split()) == 8


def test_checkpoint_resume_continues_bit_exactly(tmp_path):
    """Save after 4 updates (parameters, Adam slots, EMA shadows, step), restore into a fresh replica and
    run 3 more updates on both: identical weights, slots and shadows."""
    from zero_amd.main import Trainer, _restore
    from zero_amd.utils.saver import Saver, collect_tensors
    reset_cores(); reset_stores()
    hp = make_hp("transformer_aan", ema_decay=0.99, lrate=0.3, warmup_steps=5, scope_name="resume_a")
    rng = np.random.default_rng(2)
    batches = [make_batch(rng, 4, 6 + i, 7 + i, hp.src_vocab.size(), hp.tgt_vocab.size()) for i in range(4)]
    a = Trainer(hp)
    for i in range(4):
        a.step({"source": batches[i][0], "target": batches[i][1]})
    torch.cuda.synchronize()
    sv = Saver(output_dir=str(tmp_path))
    sv.save(collect_tensors(a.store, "resume_a", a.global_step, hp, ema=a.train_op.ema), a.global_step)
    import copy
    hp_b = copy.copy(hp); hp_b.scope_name = "resume_b"; hp_b.random_seed = 999     # different initial weights
    b = Trainer(hp_b)
    tensors = Saver(output_dir=str(tmp_path)).restore()
    # the checkpoint was written under scope resume_a: rename as a user restoring into another scope would
    tensors = {k.replace("resume_a/", "resume_b/"): v for k, v in tensors.items()}
    class _One(object):
        def restore(self, path=None):
            return tensors
    assert _restore(b, _One())
    assert b.global_step == 4 and b.store.step == 4
    for i in (1, 3, 0):
        for t in (a, b):
            t.step({"source": batches[i][0], "target": batches[i][1]})
    torch.cuda.synchronize()
    for which in ("master", "m", "v"):
        assert torch.equal(getattr(a.store, which), getattr(b.store, which)), which
    assert torch.equal(a.train_op.ema, b.train_op.ema)
