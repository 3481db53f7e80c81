#!/usr/bin/env bash
# streaming kernel at the config-3 shape (64 x 500 x 60, float32): phase timers and the kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f32 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
for v in unit global; do
timeout 120 python tools/dbg/const_timing.py 64 500 60 f32 $v 2>&1 | grep -v amdgpu.ids
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f32 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
for v in unit global; do
rm -rf gpurun_out/cprof0; rocprofv3 --kernel-trace --stats -d gpurun_out/cprof0 -o run -- python tools/dbg/const_timing.py 64 500 60 f32 $v > gpurun_out/cprof0.log 2>&1; python tools/rocpd_summary.py $(find gpurun_out/cprof0 -name "*.db" | head -1) 2>&1 | cut -c1-70,112-260 | head -6; rm -rf gpurun_out/cprof0
done
