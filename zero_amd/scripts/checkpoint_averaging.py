# coding: utf-8
"""Average the newest N checkpoints of a run (the reference's scripts/checkpoint_averaging.py) on top of
the tensor-bundle reader / writer of zero_amd/utils/bundle.py -- no TensorFlow, no GPU.

  python -m zero_amd.scripts.checkpoint_averaging --path train --checkpoints 5 --output avg

Same observable behaviour: the checkpoints listed in ``<path>/checkpoint`` (first line skipped) are
ranked by their step, the newest N that exist are averaged variable by variable in float64 and cast
back to each variable's dtype; ``global_step`` is not averaged and is written as 0; the result is
``<output>/average-0`` plus a ``checkpoint`` state file, and the ``*.json`` files of the run
(param.json, record.json) are copied next to it.
"""

import argparse
import glob
import os
import shutil

import numpy as np

from zero_amd.utils import bundle


def get_checkpoints(path):
    state = os.path.join(path, "checkpoint")
    if not os.path.exists(state):
        raise ValueError("Cannot find checkpoints in %s" % path)
    found = []
    with open(state) as fd:
        fd.readline()                      # model_checkpoint_path: the same name appears again below
        for line in fd:
            name = line.strip().split(":")[-1].strip()[1:-1]
            found.append((int(name.split("-")[-1]), os.path.join(path, name)))
    return [p for _, p in sorted(found, key=lambda kv: kv[0], reverse=True)]


def average(path, n_checkpoints, output):
    prefixes = [p for p in get_checkpoints(path)[:n_checkpoints] if os.path.exists(p + ".index")]
    if not prefixes:
        raise ValueError("None of the provided checkpoints exist. %s" % n_checkpoints)
    sums, dtypes = {}, {}
    for prefix in prefixes:
        for name, arr in bundle.load_checkpoint(prefix).items():
            if name.startswith("global_step"):
                continue
            dtypes[name] = arr.dtype
            sums[name] = sums.get(name, 0.0) + arr.astype(np.float64)
    out = {name: (total / len(prefixes)).astype(dtypes[name]) for name, total in sums.items()}
    out["global_step"] = np.array(0, dtype=np.int64)
    os.makedirs(output, exist_ok=True)
    bundle.save_checkpoint(os.path.join(output, "average-0"), out)
    open(os.path.join(output, "average-0.meta"), "wb").close()
    with open(os.path.join(output, "checkpoint"), "w") as w:
        w.write('model_checkpoint_path: "average-0"\nall_model_checkpoint_paths: "average-0"\n')
    for name in glob.glob(os.path.join(path, "*.json")):
        shutil.copyfile(name, os.path.join(output, os.path.basename(name)))
    return prefixes


def main(argv=None):
    ap = argparse.ArgumentParser(description="Average checkpoints")
    ap.add_argument("--path", type=str, required=True, help="checkpoint dir")
    ap.add_argument("--checkpoints", type=int, required=True, help="number of checkpoints to use")
    ap.add_argument("--output", type=str, required=True, help="output path")
    ap.add_argument("--gpu", type=int, default=0, help="ignored (kept for command-line compatibility)")
    a = ap.parse_args(argv)
    used = average(a.path, a.checkpoints, a.output)
    print("Averaged %d checkpoints into %s" % (len(used), os.path.join(a.output, "average-0")))


if __name__ == "__main__":
    main()
