"""Deterministic probe for scripts/launch_blocking_twin.sh: a few captured training steps (dropout on), an accumulation
cycle, and a beam-search decode, printed as exact values (float hex + CRC of the weights).  Run once with
asynchronous launches and once with HIP_LAUNCH_BLOCKING=1: any difference is an ordering bug (a missing stream
dependency that asynchrony hides or exposes)."""
import os, sys, zlib, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import make_hp, make_batch
from zero_amd.main import Trainer, tower_infer_graph
from zero_amd.models import model as registry, load_all
from zero_amd.models._factory import reset_cores

load_all()
out = {}
for model in ("transformer", "transformer_aan"):
    reset_cores()
    rng = np.random.default_rng(3)
    hp = make_hp(model, dropout=0.1, relu_dropout=0.1, residual_dropout=0.1, attention_dropout=0.1, lrate=0.5,
                 warmup_steps=10)
    src, tgt = make_batch(rng, 8, 12, 11, hp.src_vocab.size(), hp.tgt_vocab.size())
    tr = Trainer(hp)
    tr.prepare_static({"source": src, "target": tgt})
    losses = [float(tr.step_static().cpu()[0]).hex() for _ in range(6)]
    torch.cuda.synchronize()
    g, p, bad = tr.train_op.stats()
    out[model] = {"losses": losses, "gnorm": float(g).hex(), "pnorm": float(p).hex(), "bad": bad,
                  "weights_crc": zlib.crc32(tr.store.master.cpu().numpy().tobytes())}
    hp.beam_size = 4
    seqs, scores = tower_infer_graph({"source": src}, registry.get_model(model), hp)
    out[model]["beam_crc"] = zlib.crc32(np.ascontiguousarray(seqs).tobytes())
    out[model]["beam_scores_crc"] = zlib.crc32(np.ascontiguousarray(scores).tobytes())
print(json.dumps(out, indent=1, sort_keys=True))
