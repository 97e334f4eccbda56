#!/bin/bash
# MFMA-pipe utilisation of every kernel of the bench step: ONE rocprofv3 --pmc pass (SQ counters + GRBM_GUI_ACTIVE,
# with --kernel-trace only -- no other trace domain beside counters) over bench.py --steps 2 --no-graph, aggregated per
# kernel name into gpurun_out/pmc_mfma.json.  Copy that file to profiles/rNN_pmc_mfma.json: bench.py reads the newest
# one for `roofline.mfma_busy`.
cd "$(dirname "$0")/.."; export PYTHONPATH=$PWD TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf gpurun_out/pmcm
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
   -d $OLDPWD/gpurun_out/pmcm -o p --output-format csv -- \
   python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-decode --no-graph > $OLDPWD/gpurun_out/pmcm.log 2>&1); echo "pmc_mfma rc=$?"
python scripts/pmc_mfma.py gpurun_out/pmcm > gpurun_out/pmc_mfma.json
rm -rf gpurun_out/pmcm
head -c 900 gpurun_out/pmc_mfma.json; echo
