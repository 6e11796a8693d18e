#!/bin/bash
# build first; only go to the GPU box when the build is clean.  usage: scripts/gpu.sh <timeout> '<command>'
set -e
cd /root/repo
make -C conjugate-gradient-pyopencl_amd/csrc -j8 2>&1 | grep -E "error|warning: |Error" && { echo "BUILD FAILED"; exit 1; } || true
make -C conjugate-gradient-pyopencl_amd/csrc -j8 >/dev/null
make -C oracle >/dev/null
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
