#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -3
for sh in 2 3; do
  echo "=== shape $sh"
  MLPG_CONST_SHAPE=$sh timeout 300 python tools/dbg/const_time.py 2>&1 | grep -v amdgpu.ids | grep "const\|ERR" | grep "c2g f64\|c3 f32\|c5"
  MLPG_CONST_SHAPE=$sh timeout 300 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -1
done
MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
for sh in 2 3; do
  echo "=== timers, shape $sh"
  MLPG_CONST_SHAPE=$sh timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep -v amdgpu.ids
done
