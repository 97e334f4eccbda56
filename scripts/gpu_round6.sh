#!/bin/bash
# One GPU-box session of round 6.  Stages (each its own process + timeout), chosen by $1:
#   all       the full GPU suite
#   beam      the BASELINE-size decode parity tests
#   smoke     __graft_entry__.smoke()
#   bench / benchq   the default bench line / training leg only
#   profstep  rocprofv3 kernel table of the CAPTURED training step only (+ gaps)
#   profd32 / profd16   rocprofv3 kernel table of a single-lane decode, fp32 mode / bf16 mode (scripts/decode_bench.py)
#   dec32 / dec16       scripts/decode_bench.py without the profiler
#   prof                rocprofv3 kernel table of the default training bench (incl. start-up and the eager profiling steps)
#   mfma / traffic / trafficd   PMC passes (scripts/pmc_*.sh), each in runs of its own
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PYTHONPATH=$PWD TMPDIR=/tmp
STAGES=${1:-"all smoke benchq"}
prof_decode() {   # $1 dtype, $2 tag
  (cd /tmp && ZERO_HIP_DECODE_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/profd_$2 -o r1 -- python $OLDPWD/scripts/decode_bench.py --sentences 128 --dtype $1 > $OLDPWD/gpurun_out/profd_$2.log 2>&1); echo "rc=$?"
  python scripts/prof_summary.py $(find gpurun_out/profd_$2 -name "*.db" | head -1) 1 > gpurun_out/rocprof_decode_$2.txt 2>&1; rm -rf gpurun_out/profd_$2
  tail -1 gpurun_out/profd_$2.log; head -${PROF_HEAD:-40} gpurun_out/rocprof_decode_$2.txt
}
for st in $STAGES; do
  echo "=== stage $st $(date +%T)"
  case $st in
    all)    timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/all_gpu.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR" gpurun_out/all_gpu.log | head -40; tail -3 gpurun_out/all_gpu.log ;;
    beam)   timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "beam or fixture" > gpurun_out/beam.log 2>&1; echo "rc=$?"; grep -E "^FAILED|^ERROR" gpurun_out/beam.log | head; tail -4 gpurun_out/beam.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.log ;;
    bench)  timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err ;;
    benchq) timeout 600 python bench.py --no-cpu-baseline --no-decode > gpurun_out/benchq.json 2> gpurun_out/benchq.err; echo "rc=$?"; grep -o '"ms_per_step": [0-9.]*\|"static_batch_ms_per_step": [0-9.]*\|"launches_per_step": [0-9]*' gpurun_out/benchq.json; tail -3 gpurun_out/benchq.err ;;
    profstep) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/profs -o r1 -- python $OLDPWD/bench.py --steps 40 --warmup 2 --timed-only > $OLDPWD/gpurun_out/profstep.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/profs -name "*.db" | head -1) 42 > gpurun_out/rocprof_captured_step.txt 2>&1
            python scripts/prof_gaps.py $(find gpurun_out/profs -name "*.db" | head -1) > gpurun_out/rocprof_captured_step_gaps.txt 2>&1
            rm -rf gpurun_out/profs; head -${PROF_HEAD:-50} gpurun_out/rocprof_captured_step.txt; tail -2 gpurun_out/profstep.log; head -${GAP_HEAD:-12} gpurun_out/rocprof_captured_step_gaps.txt ;;
    prof)   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -o r1 -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode > $OLDPWD/gpurun_out/prof.log 2>&1); echo "rc=$?"
            python scripts/prof_summary.py $(find gpurun_out/prof -name "*.db" | head -1) > gpurun_out/rocprof_kernel_stats.txt 2>&1
            rm -rf gpurun_out/prof; head -${PROF_HEAD:-45} gpurun_out/rocprof_kernel_stats.txt ;;
    mfma)   bash scripts/pmc_mfma.sh ;;
    traffic) bash scripts/pmc_traffic.sh ;;
    trafficd) bash scripts/pmc_traffic_decode.sh ;;
    profd32) prof_decode float32 f32_1lane ;;
    profd16) prof_decode bfloat16 bf16_1lane ;;
    dec32)  ZERO_HIP_DECODE_STREAMS=1 timeout 600 python scripts/decode_bench.py --sentences 256 --dtype float32 2>&1 | tail -1 ;;
    dec16)  ZERO_HIP_DECODE_STREAMS=1 timeout 600 python scripts/decode_bench.py --sentences 256 --dtype bfloat16 2>&1 | tail -1 ;;
    decode) timeout 600 python bench.py --mode decode > gpurun_out/bench_decode.json 2> gpurun_out/bench_decode.err; echo "rc=$?"; tail -c 1200 gpurun_out/bench_decode.json ;;
    *)      echo "unknown stage $st" ;;
  esac
done
echo "=== done $(date +%T)"
