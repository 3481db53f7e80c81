#!/usr/bin/env bash
# Round 6: 128-frame strips (-DMLPG_STRIP_W=8: one workgroup of 8 wavefronts per CU) re-measured behind the 3-strip window
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_w8
: > ${O}.txt
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_w8.so timeout 900 python -m pytest tests/test_strip_gpu.py tests/test_mlpg_gpu.py -m gpu -q 2>&1 | tail -n 4 | tee -a ${O}.txt
for round in 1 2 3; do
  timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_w8.so timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
sed -i 's/R = (T + 63) \/\/ 64/R = (T + 127) \/\/ 128/' tools/dbg/strip_trace.py
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_w8trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v "amdgpu.ids\|XCD [1-7]:" | cut -c1-900 | tee -a ${O}.txt
