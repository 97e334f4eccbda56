#!/bin/bash
# Where the waves of the step's kernels spend their cycles: one rocprofv3 --pmc pass (SQ only, --kernel-trace only)
# over bench.py --steps 2 --no-graph; per-kernel sums printed as fractions of SQ_WAVE_CYCLES.
# usage: scripts/pmc_stall.sh [ENV=... ...]   (extra environment for bench.py)
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/pmcs
(cd /tmp && env "$@" timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS \
   -d $OLDPWD/gpurun_out/pmcs -o p --output-format csv -- \
   python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-decode --no-graph > $OLDPWD/gpurun_out/pmcs.log 2>&1); echo "pmc_stall rc=$?"
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmcs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
print("%-58s %10s %8s %8s %8s %9s %9s" % ("kernel", "wave_cyc", "wait_any", "wait_ins", "active", "lds_confl", "lds_idx"))
for k, c in rows:
    w = c.get("SQ_WAVE_CYCLES", 1.0) or 1.0
    print("%-58s %10.3g %8.3f %8.3f %8.3f %9.3f %9.3f" % (k[:58], w, c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w,
          c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), c.get("SQ_LDS_IDX_ACTIVE", 0) / w))
PY
rm -rf gpurun_out/pmcs
