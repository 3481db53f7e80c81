#!/usr/bin/env bash
# rocprofv3 kernel trace of the constant-coefficient path (per-kernel durations), shapes 2 and 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for sh in 2 3; do
  rm -rf gpurun_out/cprof$sh
  MLPG_CONST_SHAPE=$sh rocprofv3 --kernel-trace --stats -d gpurun_out/cprof$sh -o run -- python tools/dbg/const_timing.py 256 1000 60 f64 global > gpurun_out/cprof$sh.log 2>&1
  grep median gpurun_out/cprof$sh.log
  db=$(find gpurun_out/cprof$sh -name "*.db" | head -1)
  python tools/rocpd_summary.py "$db" 2>&1 | cut -c1-70,112-260 | head -12
  rm -rf gpurun_out/cprof$sh
done
