#!/usr/bin/env bash
# Round 6: the strip kernel's start ramp (MLPG_STRIP_STAGGER_US): sweep of the span, interleaved rounds; trace at one value
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_stagger
: > ${O}.txt
for round in 1 2; do
for us in ${SWEEP:-0 6 10 14 18 22 26 32}; do
  echo "== stagger $us us (round $round)" | tee -a ${O}.txt
  MLPG_STRIP_STAGGER_US=$us timeout 120 python tools/dbg/strip_variant_time.py ${WHAT:-fwd} both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
for us in ${TRACE:-0 18}; do
echo "== trace at $us us" | tee -a ${O}.txt
MLPG_STRIP_STAGGER_US=$us NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tee -a ${O}.txt
done
