#!/bin/bash
# HBM-side traffic of every kernel of the bench step: two rocprofv3 --pmc passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass), aggregated per kernel name into gpurun_out/pmc_traffic.json.
# Copy that file to profiles/rNN_pmc_traffic.json: bench.py reads the newest one for `traffic`.
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmct_$c
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/gpurun_out/pmct_$c -o p --output-format csv -- \
     python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-decode --no-graph > $OLDPWD/gpurun_out/pmct_$c.log 2>&1); echo "$c rc=$?"
done
python scripts/pmc_traffic.py gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE --clusters > gpurun_out/pmc_traffic.json
rm -rf gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE      # raw counter CSVs: tens of MB, gpurun merges back at most 64 MiB
head -c 600 gpurun_out/pmc_traffic.json; echo
