#!/bin/bash
# Ordering check: the same training steps + decode with asynchronous launches and with
# HIP_LAUNCH_BLOCKING=1 must print identical values; then the model / loop / data-parallel GPU tests once more under
# blocking launches.  GPU box.  Log: gpurun_out/launch_blocking_twin.log (copy to profiles/).
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
L=gpurun_out/launch_blocking_twin.log
{
  python scripts/twin_run.py > gpurun_out/twin_async.json 2>gpurun_out/twin_async.err; echo "# async run rc=$?"
  HIP_LAUNCH_BLOCKING=1 python scripts/twin_run.py > gpurun_out/twin_blocking.json 2>gpurun_out/twin_blocking.err; echo "# blocking run rc=$?"
  if diff gpurun_out/twin_async.json gpurun_out/twin_blocking.json > gpurun_out/twin.diff; then
    echo "# IDENTICAL: asynchronous and HIP_LAUNCH_BLOCKING=1 runs print the same values:"; cat gpurun_out/twin_async.json
  else
    echo "# DIFFERENT:"; cat gpurun_out/twin.diff
  fi
  echo "# GPU tests under HIP_LAUNCH_BLOCKING=1:"
  HIP_LAUNCH_BLOCKING=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_loops.py tests/test_gpu_dp.py tests/test_gpu_sync_ln.py tests/test_gpu_decode_f32.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
} > $L 2>&1
tail -8 $L
