"""Beam-search decode throughput (BASELINE config 4): transformer_aan, beam 4, alpha 0.6,
decode_length 50, eval_batch_size 32, synthetic sources with lengths ~ clipped N(28,14) in
[4,100] + eos, Transformer-base sizes, random weights (outputs rarely emit
EOS, so every batch runs to its length cap: the figure to read is decode STEPS per second).

usage: python scripts/decode_bench.py [--sentences 256] [--model transformer_aan]"""
import argparse, os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.config import transformer_base_params, SyntheticVocab
from zero_amd.models import model as registry, load_all
from zero_amd.main import tower_infer_graph

ap = argparse.ArgumentParser()
ap.add_argument("--sentences", type=int, default=256)
ap.add_argument("--model", default="transformer_aan")
ap.add_argument("--beam", type=int, default=4)
ap.add_argument("--dtype", default="bfloat16", help="decode_dtype: bfloat16 (product mode) or float32 (token-exact mode, zk_f32_*)")
args = ap.parse_args()
load_all()
V = 32000
hp = transformer_base_params(model_name=args.model, scope_name=args.model, beam_size=args.beam, decode_alpha=0.6,
                             decode_length=50, eval_batch_size=32)
hp.src_vocab = SyntheticVocab(V); hp.tgt_vocab = SyntheticVocab(V)
hp.decode_dtype = args.dtype
rng = np.random.default_rng(1234)
lens = np.clip(np.rint(rng.normal(28, 14, args.sentences)), 4, 100).astype(int)
order = np.argsort(lens, kind="stable")                     # length-sorted batches (data.py:69-73)
graph = registry.get_model(args.model)
tot_steps = tot_sent = tot_tok = 0
t_all = 0.0
for b0 in range(0, args.sentences, hp.eval_batch_size):
    idx = order[b0:b0 + hp.eval_batch_size]
    L = int(lens[idx].max()) + 1
    src = np.zeros((len(idx), L), dtype=np.int64)
    for r, i in enumerate(idx):
        src[r, :lens[i]] = rng.integers(3, V, lens[i]); src[r, lens[i]] = 2
    torch.cuda.synchronize(); t0 = time.perf_counter()
    from zero_amd.search import beam_search
    enc, dec = graph.infer_fn(hp)
    out = beam_search({"source": src}, enc, dec, hp)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if b0 > 0:                                              # first batch = warm-up (buffer sizing)
        t_all += dt; tot_steps += out["steps"]; tot_sent += len(idx)
        tot_tok += int((out["seq"][:, 0] != 0).sum())
print(json.dumps({"model": args.model, "decode_dtype": args.dtype, "beam": args.beam, "sentences": tot_sent, "decode_steps": tot_steps,
                  "seconds": t_all, "steps_per_s": tot_steps / t_all, "sentences_per_s": tot_sent / t_all,
                  "ms_per_step": 1e3 * t_all / tot_steps, "rows_per_step": hp.eval_batch_size * args.beam}))
