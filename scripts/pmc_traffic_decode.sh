#!/bin/bash
# HBM-side traffic of the decode leg (BASELINE configs[3]): two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
# bench.py --mode decode on ONE lane (counter collection serialises the dispatches anyway), summed over every kernel of the
# job and divided by the number of decode steps (= launches of k_beam_prepare, one per step) into
# gpurun_out/pmc_traffic_decode.json.  Copy it to profiles/rNN_pmc_traffic_decode.json: bench.py reads the newest one
# for the `traffic` of the decode object.
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmcd_$c
  (cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/gpurun_out/pmcd_$c -o p --output-format csv -- \
     python $OLDPWD/bench.py --mode decode --sentences 320 --decode-streams 1 --no-cpu-baseline --no-modellike > $OLDPWD/gpurun_out/pmcd_$c.log 2>&1); echo "$c rc=$?"
done
python scripts/pmc_traffic.py gpurun_out/pmcd_FETCH_SIZE gpurun_out/pmcd_WRITE_SIZE --decode > gpurun_out/pmc_traffic_decode.json
rm -rf gpurun_out/pmcd_FETCH_SIZE gpurun_out/pmcd_WRITE_SIZE
head -c 700 gpurun_out/pmc_traffic_decode.json; echo
