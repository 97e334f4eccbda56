#!/bin/bash
# Wrapper around gpurun used during development: records the commit the snapshot was taken at (the GPU box
# receives no .git) so that measurement files can say which code they belong to, then forwards the command.
# usage: scripts/gpu.sh [--timeout S] '<command>'
cd "$(dirname "$0")/.."
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet || echo +dirty)" > .head_commit
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
exec /usr/local/graft/bin/gpurun --timeout $T -- "export PYTHONPATH=\$PWD TMPDIR=/tmp; mkdir -p gpurun_out; $*"
