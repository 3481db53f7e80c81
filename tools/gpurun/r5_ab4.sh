#!/usr/bin/env bash
# Round 5: the 3-strip level-3 window (route 1) with acceptance at 1e-19 (experiment, no fallback ladder yet): interleaved A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_ab4
: > ${O}.txt
for round in 1 2 3; do
  timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_route1.so timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
