#!/usr/bin/env bash
# Round 5: the strip launch's last workgroup does verdict_kernel's work (MLPG_STRIP_FOLD): the whole suite, interleaved A/B against
# the two-launch form (tools/dbg/bin/libmlpg_hip_nofold.so), the bench line both ways, a soak.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_fold
: > ${O}.txt
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1; echo "pytest rc=$?" >> ${O}_tests.log; tail -n 4 ${O}_tests.log
NOFOLD=$PWD/tools/dbg/bin/libmlpg_hip_nofold.so
for round in 1 2 3; do
  timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$NOFOLD timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
done
for round in 1 2; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --regions 0 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold   ', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" | tee -a ${O}.txt
  NNMNKWII_AMD_SO=$NOFOLD timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --regions 0 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nofold ', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" | tee -a ${O}.txt
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r5_fold_trace -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --regions 0 > ${O}_trace.log 2>&1
f=$(find gpurun_out/r5_fold_trace -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" > ${O}_trace.txt 2>&1; rm -rf gpurun_out/r5_fold_trace
head -5 ${O}_trace.txt | cut -c1-200
timeout 200 python tools/dbg/soak_strip.py 90 41 2>&1 | tail -n 2 | tee -a ${O}.txt
