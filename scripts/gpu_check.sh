#!/bin/bash
# One GPU-box session: kernel parity, model parity, smoke, bench, rocprof.  Every stage is
# isolated (own process + timeout) and logs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD
export TMPDIR=/tmp
STAGES=${1:-"kernels model smoke bench prof"}
rocm-smi --showproductname > gpurun_out/smi.log 2>&1 || true
for st in $STAGES; do
  echo "=== stage $st $(date +%T)"
  case $st in
    kernels) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -rA --tb=short -n 1 --max-worker-restart 30 \
               -p no:cacheprovider > gpurun_out/kernels.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/kernels.log ;;
    model)   timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -rA --tb=short -n 1 --max-worker-restart 30 \
               -p no:cacheprovider > gpurun_out/model.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/model.log ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.log ;;
    bench)   timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench.log ;;
    benchng) timeout 900 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > gpurun_out/bench_nograph.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/bench_nograph.log ;;
    prof)    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/gpurun_out/prof.log 2>&1); echo "rc=$?"; tail -2 gpurun_out/prof.log
             ls gpurun_out/prof 2>/dev/null | head ;;
    pmc)     for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES"; do
               tag=$(echo $grp | cut -d' ' -f1)
               (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OLDPWD/gpurun_out/pmc_$tag -o p --output-format csv -- python $OLDPWD/scripts/gemm_pmc.py > $OLDPWD/gpurun_out/pmc_$tag.log 2>&1); echo "pmc $tag rc=$?"
             done; ls gpurun_out/pmc_* | head -20 ;;
    all)     timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/all_gpu.log ;;
  esac
done
echo "=== done $(date +%T)"
