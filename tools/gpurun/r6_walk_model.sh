#!/usr/bin/env bash
# Round 6: what would a kernel without any inter-workgroup level cost at ONE workgroup per CU?  (-DMLPG_STRIP_ABLATE=1 + MLPG_STRIP_GRID_CAP=256)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_walk_model
: > ${O}.txt
for cap in 0 256; do
for st in 0 10; do
echo "== ablate 1, grid cap $cap, ramp $st us" | tee -a ${O}.txt
MLPG_STRIP_STAGGER_US=$st MLPG_STRIP_GRID_CAP=$cap NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_abl1.so timeout 120 python tools/dbg/strip_variant_time.py fwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
MLPG_STRIP_STAGGER_US=$st MLPG_STRIP_GRID_CAP=$cap NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_abl1trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v "amdgpu.ids\|XCD [1-7]:\|loading per" | tee -a ${O}.txt
done
done
