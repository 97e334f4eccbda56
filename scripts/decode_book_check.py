"""Full-size check of the device-resident search bookkeeping: one BASELINE-config-4 batch (32 sentences, beam 4,
V = 32000, base widths, random weights) decoded with the host-C bookkeeping and with the device-resident one must give
identical hypotheses, scores and step counts.  usage: python scripts/decode_book_check.py [--model transformer_aan]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zero_amd.config import transformer_base_params, SyntheticVocab
from zero_amd.models import model as registry, load_all
from zero_amd import search

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="transformer_aan")
args = ap.parse_args()
load_all()
V = 32000
hp = transformer_base_params(model_name=args.model, scope_name=args.model, beam_size=4, decode_alpha=0.6,
                             decode_length=50, eval_batch_size=32)
hp.src_vocab = SyntheticVocab(V); hp.tgt_vocab = SyntheticVocab(V)
rng = np.random.default_rng(7)
lens = np.clip(np.rint(rng.normal(28, 10, 32)), 4, 100).astype(int)
src = np.zeros((32, int(lens.max()) + 1), dtype=np.int64)
for r in range(32):
    src[r, :lens[r]] = rng.integers(3, V, lens[r]); src[r, lens[r]] = 2
outs = {}
for name, env in (("host_c", "0"), ("device", "1")):
    os.environ["ZERO_HIP_DECODE_DEVICE_BOOK"] = env
    enc, dec = registry.get_model(args.model).infer_fn(hp)
    outs[name] = search.beam_search({"source": src}, enc, dec, hp)
    torch.cuda.synchronize()
a, b = outs["host_c"], outs["device"]
same = a["steps"] == b["steps"] and np.array_equal(a["seq"], b["seq"]) and np.array_equal(a["score"], b["score"])
print("%s: steps host %d device %d, hypotheses %s, identical: %s" % (args.model, a["steps"], b["steps"], a["seq"].shape, same))
sys.exit(0 if same else 1)
