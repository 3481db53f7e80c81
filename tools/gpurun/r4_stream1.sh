#!/usr/bin/env bash
# streaming constant-coefficient kernel: parity tests (both shapes), timings, kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for sh in 0 1; do
  echo "=== tests, shape $sh: $(MLPG_CONST_SHAPE=$sh timeout 900 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -4)"
done
for sh in 0 1; do
  echo "=== shape $sh"
  MLPG_CONST_SHAPE=$sh timeout 300 python tools/dbg/const_time.py 2>&1 | grep -v amdgpu.ids | grep "const\|ERR"
done
for sh in 0 1; do
  rm -rf gpurun_out/cprof$sh
  MLPG_CONST_SHAPE=$sh rocprofv3 --kernel-trace --stats -d gpurun_out/cprof$sh -o run -- python tools/dbg/const_timing.py 256 1000 60 f64 global > gpurun_out/cprof$sh.log 2>&1
  grep median gpurun_out/cprof$sh.log
  db=$(find gpurun_out/cprof$sh -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" 2>&1 | cut -c1-70,112-260 | head -6
  rm -rf gpurun_out/cprof$sh
done
