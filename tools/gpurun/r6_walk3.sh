#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_walk3
: > ${O}.txt
for round in 1 2; do
for cfg in "default 0 0" "default 1 10" "walkring9 1 10" "walkring12 1 10" "walkring18 1 10" "walkring12 1 0"; do
  set -- $cfg
  echo "== lib $1 walk $2 ramp $3 us" | tee -a ${O}.txt
  so=$PWD/tools/dbg/bin/libmlpg_hip_$1.so; [ "$1" = default ] && so=$PWD/nnmnkwii_amd/csrc/libmlpg_hip.so
  NNMNKWII_AMD_SO=$so MLPG_STRIP_WALK=$2 MLPG_WALK_STAGGER_US=$3 timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
done
