#!/usr/bin/env bash
# Round 6: the two workgroups of a CU taking turns at level 1 by utterance colour (MLPG_STRIP_PAIR = bounded wait in polls)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_pair
: > ${O}.txt
for round in 1 2; do
for cfg in ${SWEEP:-"0 0" "50 0" "200 0" "1000 0" "4000 0" "1000 14"}; do
  set -- $cfg
  echo "== pair $1 stagger $2 (round $round)" | tee -a ${O}.txt
  MLPG_STRIP_PAIR=$1 MLPG_STRIP_STAGGER_US=$2 timeout 120 python tools/dbg/strip_variant_time.py ${WHAT:-fwd} ${DT:-f64} 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
for cfg in ${TRACE:-"1000 0"}; do
set -- $cfg
echo "== trace at pair $1 stagger $2" | tee -a ${O}.txt
MLPG_STRIP_PAIR=$1 MLPG_STRIP_STAGGER_US=$2 NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v "amdgpu.ids\|XCD [1-7]:" | cut -c1-1300 | tee -a ${O}.txt
done
