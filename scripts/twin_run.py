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
    # round 4: the loop a user runs -- Trainer.step on ROTATING batches (next batch uploaded and prepared on a side stream
    # into a staging set, one commit launch, captured step) with no host synchronisation in between
    feats = []
    for i in range(4):
        s_, t_ = make_batch(np.random.default_rng(40 + i), 8, 12, 11, hp.src_vocab.size(), hp.tgt_vocab.size())
        s_[:, -1], t_[:, -1] = 2, 2
        feats.append({"source": s_, "target": t_})
    held = [tr.step(feats[i % 4]).reshape(-1)[0].clone() for i in range(10)]
    torch.cuda.synchronize()
    out[model]["rotating_losses"] = [float(x.cpu()).hex() for x in held]
    out[model]["rotating_weights_crc"] = zlib.crc32(tr.store.master.cpu().numpy().tobytes())
    # round 4: four decode batches in flight on execution lanes (evalu.decode_many; zk_beam_dev_run called from four host
    # threads, start-ups serialised by a lock, replays free)
    import copy, threading
    from zero_amd.evalu import decode_many
    from zero_amd.search import beam_search
    hpd = copy.copy(hp); hpd.search_mode = "cache"
    tl = threading.local()

    def work(s_):
        if not hasattr(tl, "fns"):
            tl.fns = registry.get_model(model).infer_fn(hpd)
        r = beam_search({"source": s_}, tl.fns[0], tl.fns[1], hpd)
        return zlib.crc32(np.ascontiguousarray(r["seq"]).tobytes()), zlib.crc32(np.ascontiguousarray(r["score"]).tobytes()), r["steps"]
    srcs = [make_batch(np.random.default_rng(60 + i), 3 + i % 4, 6 + 2 * i, 5, hp.src_vocab.size(), hp.tgt_vocab.size())[0]
            for i in range(12)]
    out[model]["lanes4"] = decode_many(srcs, work, streams=4)
print(json.dumps(out, indent=1, sort_keys=True))
