#!/usr/bin/env bash
# Round 6: the strip kernel with the last frames of a chunk arriving by LDS-DMA (-DMLPG_STRIP_DMA=n): parity + interleaved A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_dma
: > ${O}.txt
for tag in "$@"; do
  echo "== parity: $tag" | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$tag.so timeout 900 python -m pytest tests/test_strip_gpu.py tests/test_mlpg_gpu.py tests/test_parity_r2_gpu.py -m gpu -q -x 2>&1 | tail -n 3 | tee -a ${O}.txt
done
for round in 1 2 3; do
  timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  for tag in "$@"; do
    NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$tag.so timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  done
done
