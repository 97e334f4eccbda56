#!/bin/bash
# One GPU-box session of round 5.  Stages (each isolated: own process + timeout), chosen by $1:
#   new      the tests written / changed this round (fp32 decode mode, exchange hardening)
#   all      the full GPU suite;  allk: the same with PYTEST_K as -k expression and without -x
#   beam     the BASELINE-size decode parity tests (new fixture: bf16 criterion + fp32 token-exact)
#   smoke    __graft_entry__.smoke()
#   bench    the default bench line;  benchq: training leg only
#   profstep rocprofv3 kernel table of the CAPTURED step only (bench.py --timed-only: warm-up + timed replays, nothing else)
#   prof     kernel table of the whole default training leg (as rounds 1-4)
#   mfma / traffic / trafficd   PMC passes (scripts/pmc_*.sh)
#   ab       same-box A/B of environment switches: AB="NAME=a NAME=b ..."
#   gemmbig  scripts/gemm_big_bench.py (tile variants of the large GEMMs)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
STAGES=${1:-"new all smoke benchq"}
for st in $STAGES; do
  echo "=== stage $st $(date +%T)"
  case $st in
    new)    timeout 900 python -m pytest tests/test_gpu_decode_f32.py tests/test_gpu_sync_ln.py tests/test_gpu_update_fused.py -m gpu -q -p no:cacheprovider > gpurun_out/new_gpu.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR|Error|assert" gpurun_out/new_gpu.log | head -40; tail -4 gpurun_out/new_gpu.log ;;
    all)    timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/all_gpu.log ;;
    allk)   timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR" gpurun_out/all_gpu.log | head -40; tail -5 gpurun_out/all_gpu.log ;;
    beam)   timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "beam or fixture" > gpurun_out/beam.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR" gpurun_out/beam.log | head; tail -4 gpurun_out/beam.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.log ;;
    bench)  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err ;;
    benchq) timeout 600 python bench.py --no-cpu-baseline --no-decode > gpurun_out/benchq.json 2> gpurun_out/benchq.err; echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*\|"feed_overhead_frac": [-0-9.e]*' gpurun_out/benchq.json; tail -3 gpurun_out/benchq.err ;;
    ab)     for kv in ${AB:-}; do fn=$(echo "$kv" | tr '/=:' '___'); echo "--- $kv"; env $kv timeout 600 python bench.py --no-cpu-baseline --no-decode --steps 40 --timed-only ${AB_ARGS:-} > gpurun_out/ab_$fn.json 2> gpurun_out/ab_$fn.err; grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$fn.json; tail -1 gpurun_out/ab_$fn.err | cut -c1-200; done ;;
    profstep) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/profs -o r1 -- python $OLDPWD/bench.py --steps 40 --warmup 2 --timed-only > $OLDPWD/gpurun_out/profstep.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/profs -name "*.db" | head -1) 42 > gpurun_out/rocprof_captured_step.txt 2>&1
            python scripts/prof_gaps.py $(find gpurun_out/profs -name "*.db" | head -1) > gpurun_out/rocprof_captured_step_gaps.txt 2>&1
            rm -rf gpurun_out/profs; head -${PROF_HEAD:-50} gpurun_out/rocprof_captured_step.txt; tail -2 gpurun_out/profstep.log; head -${GAP_HEAD:-30} gpurun_out/rocprof_captured_step_gaps.txt ;;
    prof)   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $OLDPWD/gpurun_out/prof.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/prof -name "*.db" | head -1) > gpurun_out/rocprof_kernel_stats.txt 2>&1
            rm -rf gpurun_out/prof; head -${PROF_HEAD:-45} gpurun_out/rocprof_kernel_stats.txt ;;
    profd)  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/profd -o r1 -- python $OLDPWD/bench.py --mode decode --sentences 600 --no-cpu-baseline > $OLDPWD/gpurun_out/profd.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/profd -name "*.db" | head -1) 1 > gpurun_out/rocprof_decode.txt 2>&1; rm -rf gpurun_out/profd; head -40 gpurun_out/rocprof_decode.txt ;;
    mfma)   bash scripts/pmc_mfma.sh ;;
    traffic) bash scripts/pmc_traffic.sh ;;
    trafficd) bash scripts/pmc_traffic_decode.sh ;;
    gemmsched) timeout 600 python scripts/gemm_big_bench.py --sched > gpurun_out/gemm_sched.txt 2>&1; echo "rc=$?"; tail -12 gpurun_out/gemm_sched.txt ;;
    gemmbig) timeout 600 python scripts/gemm_big_bench.py > gpurun_out/gemm_big.txt 2>&1; echo "rc=$?"; tail -40 gpurun_out/gemm_big.txt ;;
    twin)   bash scripts/launch_blocking_twin.sh ;;
    soak)   timeout 600 python scripts/soak_decode.py 200 4 > gpurun_out/soak_decode.json 2> gpurun_out/soak_decode.err; echo "rc=$?"; cat gpurun_out/soak_decode.json; tail -3 gpurun_out/soak_decode.err ;;
    decode) timeout 600 python bench.py --mode decode > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "rc=$?"; tail -c 900 gpurun_out/bench_decode.json ;;
  esac
done
echo "=== done $(date +%T)"
