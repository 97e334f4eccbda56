#!/bin/bash
# Builds the TRACE=1 variant of the library (in-kernel s_memtime stamps in the gen-2 GEMM) beside the default one, as
# zero_amd/csrc/libzero_hip_trace.so.  Use: ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_trace.so python scripts/trace_sync_ln.py
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/zk_trace_build && mkdir -p /tmp/zk_trace_build
cp zero_amd/csrc/*.hip zero_amd/csrc/*.h zero_amd/csrc/Makefile /tmp/zk_trace_build/
make -C /tmp/zk_trace_build -j8 TRACE=1 ARCH=gfx950 > /tmp/zk_trace_build/build.log 2>&1 || { tail -30 /tmp/zk_trace_build/build.log; exit 1; }
cp /tmp/zk_trace_build/libzero_hip.so zero_amd/csrc/libzero_hip_trace.so
echo "built zero_amd/csrc/libzero_hip_trace.so"
