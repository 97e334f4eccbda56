#!/bin/bash
# A/B of bench.py under env-var variants inside ONE gpurun (same box): "name|ENV=... ENV=..."
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export PYTHONPATH=$PWD
for rep in 1 2; do
for v in "$@"; do
  name=${v%%|*}; envs=${v#*|}
  out=$(env $envs python bench.py --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$rep $name $(echo $out | grep -o '"ms_per_step": [0-9.]*')"
done; done
