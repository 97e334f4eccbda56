# coding: utf-8
"""Soak of evalu.decode_many: N batches of random shapes on 4 execution lanes -- shapes chosen so
that lanes keep meeting new (beam rows, source length, cache length) keys, i.e. one lane is inside its start-up (buffer
growth, pinned staging, graph capture) while the others replay -- against the one-after-the-other loop: hypotheses, scores
and step counts must be identical, batch for batch.  Prints one JSON line.

    python scripts/soak_decode.py [batches=200] [lanes=4]
"""
import copy
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402
from tests.common import make_hp, make_batch, perturb   # noqa: E402
from oracle import ref_torch as rt   # noqa: E402   (parameters only: init_params)
from zero_amd.evalu import decode_many   # noqa: E402
from zero_amd.models import model as registry, load_all   # noqa: E402
from zero_amd.models._factory import get_core   # noqa: E402
from zero_amd.search import beam_search   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 4
load_all()
out = {"batches": N, "lanes": LANES, "models": {}}
# round 5: third leg = the fp32 decode mode (decode_dtype=float32) on the same lanes
for model, ddtype in (("transformer_aan", "bfloat16"), ("transformer", "bfloat16"), ("transformer_aan", "float32")):
    hp = make_hp(model, beam_size=4, decode_length=8, decode_dtype=ddtype)
    hp = copy.copy(hp)
    hp.search_mode = "cache"
    rng = np.random.default_rng(17)
    Pn = perturb(rt.init_params(hp, model, seed=3), rng)
    Pn["tgt_embedding"] = (Pn["tgt_embedding"] * 6.0).astype(np.float32)
    get_core(hp, model, Pn)
    batches = []
    for i in range(N):
        # growing maxima (buffers are replaced now and then), 14 length buckets x 6 batch sizes, repeats in between
        b = int(rng.integers(2, 8)) + (4 if i > N // 2 else 0)
        ls = int(rng.integers(4, 40 + (30 if i > N // 3 else 0)))
        s_, _ = make_batch(rng, b, ls, 5, hp.src_vocab.size(), hp.tgt_vocab.size())
        batches.append(s_)
    graph = registry.get_model(model)
    tl = threading.local()

    def work(s_):
        if not hasattr(tl, "fns"):
            tl.fns = graph.infer_fn(hp)
        r = beam_search({"source": s_}, tl.fns[0], tl.fns[1], hp)
        return np.asarray(r["seq"]).copy(), np.asarray(r["score"]).copy(), r["steps"]
    t0 = time.time()
    seq = decode_many(batches, work, streams=1)
    t1 = time.time()
    par = decode_many(batches, work, streams=LANES)
    t2 = time.time()
    bad = [i for i, ((a, b, c), (x, y, z)) in enumerate(zip(seq, par))
           if not (np.array_equal(a, x) and np.array_equal(b, y) and c == z)]
    shapes = len(set((s_.shape[0], -(-s_.shape[1] // 8)) for s_ in batches))
    out["models"][model + ("" if ddtype == "bfloat16" else "/" + ddtype)] = {"mismatching_batches": bad, "distinct_shape_buckets": shapes,
                            "decode_steps": int(sum(c for _, _, c in seq)),
                            "sequential_s": round(t1 - t0, 2), "lanes_s": round(t2 - t1, 2)}
    assert not bad, (model, ddtype, bad[:10])
torch.cuda.synchronize()
print(json.dumps(out))
