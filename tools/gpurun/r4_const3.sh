#!/usr/bin/env bash
# constant-coefficient kernel: timings per build flag set (forward f64 rebuilt), shapes 0 and 2, then phase timers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
  echo "=== [$flags] $(timeout 300 python -m pytest tests/test_const_gpu.py -m gpu -x -q -k 'large_shape or repeat or slow' 2>&1 | tail -1)"
  for sh in 0 2; do
    MLPG_CONST_SHAPE=$sh timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep "median" | sed "s/^/shape $sh: /"
  done
done
MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
MLPG_CONST_SHAPE=2 timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep -v amdgpu.ids
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
