#!/usr/bin/env bash
# Round 6: load-phase tokens per XCD (MLPG_STRIP_THROTTLE = workgroups of an XCD allowed in level 1 at a time), with and without the start ramp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_throttle
: > ${O}.txt
for round in 1 2; do
for cfg in ${SWEEP:-"0 0" "24 0" "28 0" "32 0" "40 0" "48 0" "32 18" "40 18"}; do
  set -- $cfg
  echo "== throttle $1 stagger $2 (round $round)" | tee -a ${O}.txt
  MLPG_STRIP_THROTTLE=$1 MLPG_STRIP_STAGGER_US=$2 timeout 120 python tools/dbg/strip_variant_time.py fwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
for cfg in ${TRACE:-"32 0" "40 18"}; do
set -- $cfg
echo "== trace at throttle $1 stagger $2" | tee -a ${O}.txt
MLPG_STRIP_THROTTLE=$1 MLPG_STRIP_STAGGER_US=$2 NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v amdgpu.ids | cut -c1-1300 | tee -a ${O}.txt
done
