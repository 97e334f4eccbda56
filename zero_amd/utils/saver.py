# coding: utf-8
"""Checkpoint rotation with the reference's directory layout (utils/saver.py:11-171), payload in the
TensorFlow V2 bundle format (zero_amd/utils/bundle.py) under the reference's variable names.

Layout kept from the reference:
  ``<output_dir>/model-<step>.{index,data-00000-of-00001,meta}`` + ``checkpoint`` (first line
  ``model_checkpoint_path: "model-<step>"`` then one ``all_model_checkpoint_paths`` line per kept
  checkpoint -- the file ``Saver.restore`` parses at utils/saver.py:138-141), at most
  ``checkpoints`` kept; ``<output_dir>/best/`` with the ``best_checkpoints`` highest-scoring
  checkpoints, ``metric.log`` (``Steps N, Metric Score S`` lines; the last one seeds
  ``best_score``, saver.py:43-49) and ``topk_checkpoint`` (``name<TAB>score``, saver.py:52-59,121-127);
  a new best also copies ``param.json`` / ``record.json`` into ``best/`` (saver.py:84-90).

Variable names inside a checkpoint: ``<scope_name>/<variable>`` for the parameters, the TF1
``AdamOptimizer`` slot names ``<scope>/<variable>/Adam`` (m) and ``.../Adam_1`` (v),
``beta1_power`` / ``beta2_power`` and int64 ``global_step`` -- what ``tf.train.Saver()`` stores
for the reference graph, so ``restore`` accepts checkpoints written by either side.  Variables
absent from a checkpoint are reported and left at their current value, the name-matching
fallback of saver.py:150-170.  The ``.meta`` file is an empty placeholder (the reference only
tests that it exists, saver.py:145).
"""

import logging
import os
import shutil

import numpy as np

from zero_amd.utils import bundle

log = logging.getLogger("zero_amd")


def _read_state(directory):
    """(latest, [all]) checkpoint names listed in ``<directory>/checkpoint``."""
    path = os.path.join(directory, "checkpoint")
    latest, every = None, []
    if os.path.exists(path):
        for line in open(path):
            if ":" not in line:
                continue
            key, val = line.split(":", 1)
            val = val.strip().strip('"')
            if key.strip() == "model_checkpoint_path":
                latest = val
            elif key.strip() == "all_model_checkpoint_paths":
                every.append(val)
    return latest, every


def _write_state(directory, names):
    with open(os.path.join(directory, "checkpoint"), "w") as w:
        if names:
            w.write('model_checkpoint_path: "{}"\n'.format(names[-1]))
        for n in names:
            w.write('all_model_checkpoint_paths: "{}"\n'.format(n))


def _remove(directory, name):
    for suffix in (".index", ".data-00000-of-00001", ".meta"):
        p = os.path.join(directory, name + suffix)
        if os.path.exists(p):
            os.remove(p)


def collect_tensors(store, scope, global_step, hparams=None, ema=None):
    """The checkpoint content of one replica: parameters, Adam slots, step, and (``ema``: the flat shadow
    buffer of TrainOp) the ExponentialMovingAverage shadows under TF's slot name."""
    out = {}
    if ema is not None:
        for name, arr in store.export(ema).items():
            out["%s/%s/ExponentialMovingAverage" % (scope, name)] = arr
    for which, suffix in (("master", ""), ("m", "/Adam"), ("v", "/Adam_1")):
        for name, arr in store.export(which).items():
            out["%s/%s%s" % (scope, name, suffix)] = arr
    out["global_step"] = np.array(int(global_step), dtype=np.int64)
    if hparams is not None:
        out["beta1_power"] = np.array(hparams.beta1 ** (store.step + 1), dtype=np.float32)
        out["beta2_power"] = np.array(hparams.beta2 ** (store.step + 1), dtype=np.float32)
    return out


def assign_flat(store, flat, scope, tensors, suffix):
    """Fill a flat buffer laid out like the parameters from ``<scope>/<name><suffix>`` entries."""
    import torch
    n = 0
    for name in store.names():
        key = "%s/%s%s" % (scope, name, suffix)
        if key in tensors and tuple(tensors[key].shape) == store.lshape[name]:
            dst = store._view(flat, name)
            t = torch.as_tensor(np.asarray(tensors[key], dtype=np.float32))
            dst.zero_()
            if t.dim() == 2:
                dst[:t.shape[0], :t.shape[1]].copy_(t)
            else:
                dst.copy_(t)
            n += 1
    return n


def assign_tensors(store, scope, tensors):
    """Name-matching restore (saver.py:150-170) -> (restored names, missing names, global_step)."""
    values, slots = store.export("master"), {"m": store.export("m"), "v": store.export("v")}
    got, missing = [], []
    for name in list(values):
        key = "%s/%s" % (scope, name)
        if key in tensors and tuple(tensors[key].shape) == tuple(values[name].shape):
            values[name] = tensors[key]
            got.append(name)
        else:
            missing.append(name)
        for which, suffix in (("m", "/Adam"), ("v", "/Adam_1")):
            k2 = key + suffix
            if k2 in tensors and tuple(tensors[k2].shape) == tuple(values[name].shape):
                slots[which][name] = tensors[k2]
    store.load(values)
    for which in ("m", "v"):
        flat = getattr(store, which)
        for name, arr in slots[which].items():
            dst = store._view(flat, name)
            dst.zero_()
            import torch
            t = torch.as_tensor(np.asarray(arr, dtype=np.float32))
            if t.dim() == 2:
                dst[:t.shape[0], :t.shape[1]].copy_(t)
            else:
                dst.copy_(t)
    step = int(tensors["global_step"]) if "global_step" in tensors else None
    return got, missing, step


class Saver(object):
    def __init__(self, checkpoints=5, output_dir=None, best_score=-1, best_checkpoints=1):
        self.output_dir = output_dir if output_dir is not None else "./output"
        self.output_best_dir = os.path.join(self.output_dir, "best")
        self.checkpoints = checkpoints
        self.best_checkpoints = best_checkpoints
        self.best_score = best_score
        _, self.kept = _read_state(self.output_dir)
        metric_log = os.path.join(self.output_best_dir, "metric.log")
        if os.path.exists(metric_log):
            lines = open(metric_log).readlines()
            if lines:
                self.best_score = float(lines[-1].strip().split()[-1])
        self.topk_scores = []
        topk = os.path.join(self.output_best_dir, "topk_checkpoint")
        if os.path.exists(topk):
            for line in open(topk):
                name, score = line.strip().split("\t")
                self.topk_scores.append((name, float(score)))
        else:
            latest, _ = _read_state(self.output_best_dir)
            if latest is not None:
                self.topk_scores.append((latest, self.best_score))

    def _dump(self, directory, step, tensors):
        os.makedirs(directory, exist_ok=True)
        name = "model-{}".format(int(step))
        bundle.save_checkpoint(os.path.join(directory, name), tensors)
        open(os.path.join(directory, name + ".meta"), "wb").close()
        return name

    def save(self, tensors, step, metric_score=None):
        """``tensors``: {checkpoint variable name: array} (see :func:`collect_tensors`)."""
        name = self._dump(self.output_dir, step, tensors)
        self.kept = [n for n in self.kept if n != name] + [name]
        while self.checkpoints and len(self.kept) > self.checkpoints:
            _remove(self.output_dir, self.kept.pop(0))
        _write_state(self.output_dir, self.kept)
        os.makedirs(self.output_best_dir, exist_ok=True)
        if metric_score is None:
            return
        if metric_score > self.best_score:
            self.best_score = metric_score
            for f in ("param.json", "record.json"):
                src = os.path.join(self.output_dir, f)
                if os.path.exists(src):
                    shutil.copyfile(src, os.path.join(self.output_best_dir, f))
            with open(os.path.join(self.output_best_dir, "metric.log"), "a+") as w:
                w.write("Steps {}, Metric Score {}\n".format(step, metric_score))
        scores = [v[1] for v in self.topk_scores]
        if not scores or len(scores) < self.best_checkpoints or metric_score > min(scores):
            self._dump(self.output_best_dir, step, tensors)
            # the same step saved again replaces its entry; whatever falls out of the top k is deleted,
            # the checkpoint just written included (ties rank by insertion order)
            self.topk_scores = [v for v in self.topk_scores if v[0] != name] + [(name, float(metric_score))]
            ranked = sorted(self.topk_scores, key=lambda x: x[1])
            for gone, _ in ranked[:-self.best_checkpoints]:
                _remove(self.output_best_dir, gone)
            self.topk_scores = ranked[-self.best_checkpoints:]
            _write_state(self.output_best_dir, [n for n, _ in self.topk_scores])
            with open(os.path.join(self.output_best_dir, "topk_checkpoint"), "w") as w:
                for n, sc in self.topk_scores:
                    w.write("{}\t{}\n".format(n, sc))

    def restore(self, path=None):
        """{name: array} of the latest checkpoint in ``path`` (or the output directory); None when
        there is none (``No Existing Model detected``, saver.py:136-137)."""
        check_dir = path if path is not None and os.path.exists(path) else self.output_dir
        latest, _ = _read_state(check_dir)
        if latest is None:
            log.warning("No Existing Model detected")
            return None
        prefix = os.path.abspath(os.path.join(check_dir, latest))
        if not os.path.exists(prefix + ".index"):
            log.error("model '%s' does not exists", prefix)
            return None
        return bundle.load_checkpoint(prefix)
