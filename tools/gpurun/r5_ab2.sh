#!/usr/bin/env bash
# Round 5: small A/B pass (interleaved, two rounds each): forward coefficients by placed scalar loads; float64 backward with
# 0 / 8 (default) / 10 early variance rows; kernel trace of the config-5 global-variance call (what the merged launch and the piece take).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_ab2
: > ${O}.txt
for round in 1 2; do
  timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_fwdkarg.so timeout 120 python tools/dbg/strip_variant_time.py fwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
for round in 1 2; do
  for v in default bwd_e0 bwd_e10; do
    if [ $v = default ]; then timeout 120 python tools/dbg/strip_variant_time.py bwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
    else NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$v.so timeout 120 python tools/dbg/strip_variant_time.py bwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt; fi
  done
done
rocprofv3 --kernel-trace --stats -d ${O}_c5 -o run -- python tools/bench_paths.py --only c5q > ${O}_c5.log 2>&1
f=$(find ${O}_c5 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > ${O}_c5.txt 2>&1; rm -rf ${O}_c5
head -12 ${O}_c5.txt | cut -c1-260
