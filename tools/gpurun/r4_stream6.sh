#!/usr/bin/env bash
# streaming kernel: chunk length 16 / 24 / 32 (MLPG_CONST_SHAPE 0 / 1 / 2), parity + timings
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for sh in 0 1 2; do
  echo "=== shape $sh: $(MLPG_CONST_SHAPE=$sh timeout 600 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -1)"
  MLPG_CONST_SHAPE=$sh timeout 300 python tools/dbg/const_time.py 2>&1 | grep -v amdgpu.ids | grep "const\|ERR" | grep "c2g f64\|c3 f32\|c5\|c2g f32"
done
