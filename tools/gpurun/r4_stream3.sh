#!/usr/bin/env bash
# streaming constant-coefficient kernel A/B: per flag set rebuild the four instantiation units, parity tests, timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_ > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
  echo "=== [$flags] $(timeout 600 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -1)"
  timeout 300 python tools/dbg/const_time.py 2>&1 | grep -v amdgpu.ids | grep "const\|ERR" | grep "c2g f64\|c3 f32\|c5\|c2g f32"
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_ > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
