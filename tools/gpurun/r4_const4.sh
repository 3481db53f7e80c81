#!/usr/bin/env bash
# constant-coefficient kernel: phase timers per build flag set and shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING $flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
  for sh in 0 2; do
    echo "=== [$flags] shape $sh"
    MLPG_CONST_SHAPE=$sh timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep -v amdgpu.ids
  done
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
