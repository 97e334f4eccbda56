#!/bin/bash
# One GPU-box session of round 4: full GPU suite, smoke, the default bench line (training + decode + cpu_baseline),
# kernel table, MFMA counters, the 256-sentence side measurement.  Each stage isolated (own process + timeout).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
STAGES=${1:-"all smoke bench prof mfma side"}
for st in $STAGES; do
  echo "=== stage $st $(date +%T)"
  case $st in
    all)    timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/all_gpu.log ;;
    allk)   timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR" gpurun_out/all_gpu.log | head -40; tail -5 gpurun_out/all_gpu.log ;;
    dp)     timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > gpurun_out/dp.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/dp.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.log ;;
    bench)  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err ;;
    benchq) timeout 600 python bench.py --no-cpu-baseline --no-decode > gpurun_out/benchq.json 2> gpurun_out/benchq.err; echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*\|"feed_overhead_frac": [-0-9.e]*' gpurun_out/benchq.json; tail -3 gpurun_out/benchq.err ;;
    ab)     # same-box A/B of environment switches: AB="NAME=a NAME=b ..." each run as one benchq
            for kv in ${AB:-}; do echo "--- $kv"; env $kv timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 30 > gpurun_out/ab_$kv.json 2> gpurun_out/ab_$kv.err; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*' gpurun_out/ab_$kv.json; tail -2 gpurun_out/ab_$kv.err; done ;;
    prof)   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $OLDPWD/gpurun_out/prof.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/prof -name "*.db" | head -1) > gpurun_out/rocprof_kernel_stats.txt 2>&1
            python scripts/prof_overlap.py $(find gpurun_out/prof -name "*.db" | head -1) ${OVERLAP:-k_batch_prep} > gpurun_out/prof_overlap.txt 2>&1; tail -12 gpurun_out/prof_overlap.txt
            rm -rf gpurun_out/prof; head -${PROF_HEAD:-45} gpurun_out/rocprof_kernel_stats.txt ;;
    profd)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/profd -o r1 -- python $OLDPWD/bench.py --mode decode --sentences 600 --no-cpu-baseline > $OLDPWD/gpurun_out/profd.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/profd -name "*.db" | head -1) 1 > gpurun_out/rocprof_decode.txt 2>&1; rm -rf gpurun_out/profd; head -40 gpurun_out/rocprof_decode.txt ;;
    mfma)   bash scripts/pmc_mfma.sh ;;
    traffic) bash scripts/pmc_traffic.sh ;;
    side)   timeout 600 python bench.py --sentences-per-gpu 256 --no-cpu-baseline --no-decode > gpurun_out/bench_b256.json 2> gpurun_out/bench_b256.err; echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"step_mfma_frac": [0-9.]*' gpurun_out/bench_b256.json ;;
    decode) timeout 600 python bench.py --mode decode > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "rc=$?"; tail -c 900 gpurun_out/bench_decode.json ;;
  esac
done
echo "=== done $(date +%T)"
