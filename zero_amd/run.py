# coding: utf-8
"""CLI contract of the reference's run.py on the HIP hot path.

Same flags (run.py:241-246: --config, --parameters, --name, --mode, --ensemble_dirs), same
parameter priority -- command line > saved ``param.json`` > ``--config`` dict file >
defaults (run.py:367-376) -- same ``param.json`` one-line JSON (run.py:250-272) and
``record.json`` (run.py:275-293).  With ``src_train_file`` / ``src_test_file`` set the modes run
the reference's loops over real bitext (zero_amd/main.py train / evaluate / scorer: data feed,
save / eval cadence, BLEU, early stopping); without data files they fall back to the synthetic
id generator so the step can be exercised on a bare GPU box.  Model ensembling
(``--mode ensemble``) is not part of the hot path and is not provided.
"""

import argparse
import ast
import json
import os
import random

import numpy as np

from zero_amd.config import default_params, SyntheticVocab
from zero_amd.utils import dtype
from zero_amd.vocab import Vocab


def save_parameters(params, output_dir):
    """run.py:250-272.  With data parallelism every process runs this CLI: only rank 0 writes, through a temporary
    file + rename, so that no other rank (or a restarted job) ever reads a torn param.json."""
    from zero_amd.utils import parallel
    if parallel.rank() != 0:
        return
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, "param.json")
    tmp = "%s.tmp.%d" % (path, os.getpid())
    with open(tmp, "w") as writer:
        writer.write(params.to_json())
    os.replace(tmp, path)


def load_parameters(params, output_dir):
    name = os.path.abspath(os.path.join(output_dir, "param.json"))
    if os.path.exists(name):
        with open(name, "r") as reader:
            params.parse_json(reader.readline())
    return params


def _safe_value(node, names):
    """Value of one config expression WITHOUT evaluating code: literals, lists / tuples / dicts of them, arithmetic
    (+ - * / // % **, unary + - not), comparisons-free boolean and / or, and names that refer to keywords defined
    EARLIER in the same ``dict(...)`` call (``hidden_size * 4``).  The reference eval()s the file (run.py:371), so
    configs with such expressions keep working; anything else (calls, attributes, subscripts, imports) is refused."""
    import operator as op
    bin_ops = {ast.Add: op.add, ast.Sub: op.sub, ast.Mult: op.mul, ast.Div: op.truediv, ast.FloorDiv: op.floordiv,
               ast.Mod: op.mod, ast.Pow: op.pow}
    if isinstance(node, ast.Constant):
        return node.value
    if isinstance(node, (ast.List, ast.Tuple)):
        vals = [_safe_value(e, names) for e in node.elts]
        return vals if isinstance(node, ast.List) else tuple(vals)
    if isinstance(node, ast.Dict):
        return {_safe_value(k, names): _safe_value(v, names) for k, v in zip(node.keys, node.values)}
    if isinstance(node, ast.Name):
        if node.id in names:
            return names[node.id]
        raise ValueError("--config: unknown name %r (only keywords defined earlier in the dict may be used)" % node.id)
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.UAdd, ast.USub, ast.Not)):
        v = _safe_value(node.operand, names)
        return +v if isinstance(node.op, ast.UAdd) else (-v if isinstance(node.op, ast.USub) else (not v))
    if isinstance(node, ast.BinOp) and type(node.op) in bin_ops:
        a, b = _safe_value(node.left, names), _safe_value(node.right, names)
        if isinstance(node.op, ast.Pow) and isinstance(b, (int, float)) and abs(b) > 64:
            raise ValueError("--config: exponent too large")
        return bin_ops[type(node.op)](a, b)
    if isinstance(node, ast.BoolOp):
        vals = [_safe_value(v, names) for v in node.values]
        out = vals[0]
        for v in vals[1:]:
            out = (out and v) if isinstance(node.op, ast.And) else (out or v)
        return out
    raise ValueError("--config: unsupported expression %s" % ast.dump(node)[:80])


def _parse_dict_call(text):
    node = ast.parse(text.strip(), mode="eval").body
    if isinstance(node, ast.Dict):
        return _safe_value(node, {})
    if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id == "dict"
            and not node.args):
        raise ValueError("--config file must hold a dict literal or a dict(k=v, ...) expression")
    out = {}
    for kw in node.keywords:
        if kw.arg is None:
            raise ValueError("--config: **kwargs are not supported")
        out[kw.arg] = _safe_value(kw.value, out)
    return out


def build_params(parameters="", config=""):
    """run.py:367-376."""
    params = default_params()
    params.parse(parameters)
    cfg = None
    if config and os.path.exists(config):
        text = open(config).read()
        try:
            cfg = ast.literal_eval(text.strip())
        except (ValueError, SyntaxError):
            # the reference eval()s a ``dict(k=v, ...)`` expression (run.py:371); accept that shape -- one call of
            # ``dict`` (or a dict display) whose values are literals and arithmetic over them -- without evaluating code
            cfg = _parse_dict_call(text)
        params.override_from_dict(cfg)
    if params.output_dir:
        params = load_parameters(params, params.output_dir)
    if cfg is not None:
        params.override_from_dict(cfg)
    params.parse(parameters)
    return params


def setup(params, synthetic_vocab=32000):
    random.seed(params.random_seed)
    np.random.seed(params.random_seed)
    params.src_vocab = Vocab(params.src_vocab_file) if params.src_vocab_file else SyntheticVocab(synthetic_vocab)
    params.tgt_vocab = Vocab(params.tgt_vocab_file) if params.tgt_vocab_file else SyntheticVocab(synthetic_vocab)
    dtype.set_floatx(params.default_dtype)
    dtype.set_epsilon(params.dtype_epsilon)
    dtype.set_inf(params.dtype_inf)
    return params


def setup_recorder(params):
    """run.py:275-293."""
    from zero_amd.utils.recorder import new_recorder
    recorder = new_recorder(params)
    path = os.path.abspath(os.path.join(params.output_dir or ".", "record.json"))
    if os.path.exists(path):
        recorder.load_from_json(path)
    if hasattr(params, "recorder"):
        params.recorder = recorder
    else:
        params.add_hparam("recorder", recorder)
    return params


def synthetic_batches(params, n_batches, sentences=64, length=64):
    rng = np.random.default_rng(params.random_seed)
    for _ in range(n_batches):
        src = rng.integers(3, params.src_vocab.size(), size=(sentences, length))
        tgt = rng.integers(3, params.tgt_vocab.size(), size=(sentences, length))
        src[:, -1] = params.src_vocab.eos()
        tgt[:, -1] = params.tgt_vocab.eos()
        yield {"source": src, "target": tgt}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="")
    ap.add_argument("--parameters", default="")
    ap.add_argument("--ensemble_dirs", default="")
    ap.add_argument("--name", default="model")
    ap.add_argument("--mode", default="train")
    args = ap.parse_args(argv)
    if "GPU_MAX_HW_QUEUES" not in os.environ:
        # this module is the host PROGRAM (the counterpart of the reference's run.py), so the process environment is
        # its to set, and it does so out loud: evaluation decodes several batches at once on execution lanes, one HIP
        # stream each, and on the runtime's default of 4 hardware queues lanes that share a queue serialise
        # (zero_amd/evalu.py decode_many).  Read by the HIP runtime at its first call -- nothing has touched it yet.
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        print("zero_amd.run: GPU_MAX_HW_QUEUES=8 exported for this process (unset; see zero_amd/evalu.py decode_many)")
    params = setup(build_params(args.parameters, args.config))
    from zero_amd.main import Trainer, tower_infer_graph, tower_score_graph
    from zero_amd.models import model, load_all
    load_all()
    import logging
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    from zero_amd import main as loops
    from zero_amd.utils import parallel
    parallel.init_distributed()
    if args.mode == "train" and params.src_train_file:
        save_parameters(params, params.output_dir or ".")
        parallel.barrier()           # nobody reads param.json / record.json before rank 0 has written them
        setup_recorder(params)
        print(json.dumps({"best_score": loops.train(params)}))
    elif args.mode == "test" and params.src_test_file:
        print(json.dumps({"bleu": loops.evaluate(params)}))
    elif args.mode == "score" and params.src_test_file:
        print(json.dumps({"score": loops.scorer(params)}))
    elif args.mode == "train":
        if params.output_dir:
            save_parameters(params, params.output_dir)
        tr = Trainer(params)
        for i, feats in enumerate(synthetic_batches(params, params.max_training_steps * params.update_cycle)):
            loss = tr.micro_step(feats)
            if (i + 1) % max(params.disp_freq, 1) == 0:
                g, p, bad = tr.train_op.stats()
                print(json.dumps({"step": tr.global_step, "loss": float(loss.cpu()), "gnorm": g, "pnorm": p}))
                if bad:
                    raise FloatingPointError("Encounter NAN or INF ERROR")
    elif args.mode == "score":
        for feats in synthetic_batches(params, 1):
            print(tower_score_graph(feats, model.get_model(params.model_name), params).cpu().numpy())
    elif args.mode == "test":
        for feats in synthetic_batches(params, 1, sentences=params.eval_batch_size, length=24):
            seqs, scores = tower_infer_graph(feats, model.get_model(params.model_name), params)
            print(seqs[:, 0], scores[:, 0])
    else:
        raise ValueError("Invalid mode: {}".format(args.mode))


if __name__ == "__main__":
    main()
