#!/bin/bash
# Builds the EXPERIMENTS=1 variant of the library (negative results kept as evidence) beside the default one, as
# zero_amd/csrc/libzero_hip_exp.so, without disturbing the default build's objects.  Use: ZERO_HIP_LIB=$PWD/zero_amd/csrc/libzero_hip_exp.so
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/zk_exp_build && mkdir -p /tmp/zk_exp_build
cp zero_amd/csrc/*.hip zero_amd/csrc/*.h zero_amd/csrc/Makefile /tmp/zk_exp_build/
make -C /tmp/zk_exp_build -j8 EXPERIMENTS=1 ARCH=gfx950 > /tmp/zk_exp_build/build.log 2>&1 || { tail -30 /tmp/zk_exp_build/build.log; exit 1; }
cp /tmp/zk_exp_build/libzero_hip.so zero_amd/csrc/libzero_hip_exp.so
echo "built zero_amd/csrc/libzero_hip_exp.so"
