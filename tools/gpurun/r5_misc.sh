#!/usr/bin/env bash
# Round 5: what a Python autograd node costs in the eager config-3 loop; AUTO's routing table; strip-kernel store variants.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_misc
timeout 300 python tools/dbg/c3_function_overhead.py 2>&1 | grep -v amdgpu.ids | tee ${O}_c3_function_overhead.txt
echo "== store variants"
timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee ${O}_stores_ab.txt
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_nt1.so timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}_stores_ab.txt
for v in bs1 bs1nt; do
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$v.so timeout 120 python tools/dbg/strip_variant_time.py bwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}_stores_ab.txt
done
echo "== strip tests on the buffer-store build"
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_bs1nt.so timeout 600 python -m pytest tests/test_strip_gpu.py tests/test_parity_r2_gpu.py tests/test_mlpg_gpu.py -m gpu -q -x 2>&1 | tail -n 3
echo "== AUTO routing"
timeout 900 python tools/auto_routing.py > ${O}_auto_routing.log 2>&1; echo "rc=$?"; tail -n 32 ${O}_auto_routing.log | cut -c1-330
