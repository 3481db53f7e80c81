#!/usr/bin/env bash
# streaming const kernel: build-flag experiments (one argument per build), parity + timings, then the T sweep of the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for t in "$@"; do
  MLPG_HIP_EXTRA_FLAGS="$t" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
  echo "=== [$t]: $(timeout 600 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -1)"
  timeout 300 python tools/dbg/const_time.py 2>&1 | grep -v amdgpu.ids | grep "const\|ERR" | grep "c2g f64\|c5\|c3 f32"
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }

