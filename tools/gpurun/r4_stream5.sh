#!/usr/bin/env bash
# streaming kernel: phase timers (forward f64, config-2 shape, global variances), then the kernel trace of the default build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for flags in "$@"; do
MLPG_HIP_EXTRA_FLAGS="-DMLPG_CONST_TIMING $flags" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
echo "=== timers [$flags]"
timeout 120 python tools/dbg/const_timing.py 256 1000 60 f64 global 2>&1 | grep -v amdgpu.ids
done
MLPG_HIP_EXTRA_FLAGS="" python nnmnkwii_amd/csrc/build.py --quiet --only=mlpg_const_fwd_f64 > /tmp/build.log 2>&1 || { echo BUILD FAILED; tail -5 /tmp/build.log; }
rm -rf gpurun_out/cprof0; rocprofv3 --kernel-trace --stats -d gpurun_out/cprof0 -o run -- python tools/dbg/const_timing.py 256 1000 60 f64 global > gpurun_out/cprof0.log 2>&1; python tools/rocpd_summary.py $(find gpurun_out/cprof0 -name "*.db" | head -1) 2>&1 | cut -c1-70,112-260 | head -4; rm -rf gpurun_out/cprof0
