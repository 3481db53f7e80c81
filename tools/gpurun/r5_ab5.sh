#!/usr/bin/env bash
# Round 5: the LDS halo hand-over (66 instead of 72 frames read per strip) re-measured behind the 3-strip window: interleaved A/B + parity
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_ab5
: > ${O}.txt
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_halo.so timeout 600 python -m pytest tests/test_strip_gpu.py tests/test_mlpg_gpu.py -m gpu -q 2>&1 | tail -n 2
for round in 1 2 3; do
  timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_halo.so timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
