#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_walk2
: > ${O}.txt
for round in 1 2; do
for cfg in "0 0" "1 0" "1 4" "1 8" "1 12" "1 16" "1 24"; do
  set -- $cfg
  echo "== walk $1 ramp $2 us" | tee -a ${O}.txt
  MLPG_STRIP_WALK=$1 MLPG_WALK_STAGGER_US=$2 timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
