#!/bin/bash
# AddressSanitizer run of the HOST side of libzero_hip.so: argument checking, error plumbing,
# the layer-program recorder, CRC32C, the beam-search host bookkeeping -- everything the CPU test-suite reaches
# through the C-ABI.  Builds an instrumented copy of the library under /tmp (device code unchanged), preloads the
# ASan runtime into python and runs the `not gpu` tests that load the library.  Log: profiles/rNN_asan_host.log.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05_asan_host.log}
B=/tmp/zk_asan; rm -rf $B; mkdir -p $B
HIPCC=/opt/rocm/bin/hipcc
for f in zk_elem zk_gemm zk_gemm2 zk_attn zk_decode zk_probe zk_comm zk_decfuse zk_rows zk_prep zk_f32; do
  $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Xarch_host -fsanitize=address \
     -Xarch_host -fno-omit-frame-pointer -c zero_amd/csrc/$f.hip -o $B/$f.o || exit 1 &
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address -o $B/libzero_hip.so $B/*.o -ldl || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(gcc -print-file-name=libasan.so)
{
  echo "# ASan runtime: $RT"
  echo "# library: $B/libzero_hip.so (host code instrumented), $(date -u +%F)"
  ZERO_HIP_LIB=$B/libzero_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 \
    python -m pytest tests/test_abi.py tests/test_host_logic.py tests/test_host_rows.py tests/test_data.py tests/test_scripts.py -m "not gpu" -q -p no:cacheprovider 2>&1 | tail -15
  echo "# exit code: ${PIPESTATUS[0]}"
  if python -c "import torch,sys; sys.exit(0 if torch.cuda.is_available() else 1)" 2>/dev/null; then
    # round 4, on the GPU box: the multi-threaded host C of the decode lanes (zk_beam_dev_run from four host threads, the
    # start-up lock, graph adoption) and the step's new host paths (zk_batch_prep / zk_copy_many argument handling)
    echo "# GPU box: decode lanes + rotating-batch steps with the instrumented host code"
    ZERO_HIP_LIB=$B/libzero_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 \
      timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider \
        -k "batches_in_flight or step_graphs_are_reused or rotating_batches or device_resident_search" 2>&1 | tail -8
    # round 5: the fp32 decode mode (zk_f32_* argument handling, 4-byte cache rows) and the give-up / skip-word paths of the exchange
    ZERO_HIP_LIB=$B/libzero_hip.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:protect_shadow_gap=0 \
      timeout 900 python -m pytest tests/test_gpu_decode_f32.py tests/test_gpu_sync_ln.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
    echo "# exit code: ${PIPESTATUS[0]}"
  fi
} > $OUT 2>&1
tail -5 $OUT
