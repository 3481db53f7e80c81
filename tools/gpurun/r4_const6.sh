#!/usr/bin/env bash
# constant-coefficient kernel A/B: per flag set: parity tests (forward f64 rebuilt), time on shapes 2 and 3; then timers
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
  echo "=== [$flags] $(MLPG_CONST_SHAPE=2 timeout 300 python -m pytest tests/test_const_gpu.py -m gpu -x -q -k 'large_shape or repeat or slow or negative' 2>&1 | tail -1)"
  for sh in 2 3; do
    for rep in 1 2; do MLPG_CONST_SHAPE=$sh timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep "median" | sed "s/^/shape $sh: /"; done
  done
done
for flags in "$@"; do
MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING $flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
echo "=== timers [$flags]"
MLPG_CONST_SHAPE=2 timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep -v amdgpu.ids
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
