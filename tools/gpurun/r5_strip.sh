#!/usr/bin/env bash
# Round 5: the strip kernel after the three-window split / backward changes (parity, soak, A/B against variant builds in
# tools/dbg/bin/), and the 8-frame-chunk experiment (MLPG_STRIP_M=8, three workgroups per CU; full and without level 3).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_strip
sum() { f=$(find ${O}_$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > ${O}_$1.txt 2>&1; rm -rf ${O}_$1; }
timeout 900 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1; echo "pytest rc=$?" >> ${O}_tests.log; tail -n 3 ${O}_tests.log
echo "== soak, strip kernel vs natural-order (60 s)"
timeout 200 python tools/dbg/soak_strip.py 60 5 > ${O}_soak.log 2>&1; echo "rc=$?" >> ${O}_soak.log; tail -n 3 ${O}_soak.log
echo "== A/B"
timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee ${O}_ab.txt
for v in bwd_old bwd_e0 bwd_e10; do
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$v.so timeout 120 python tools/dbg/strip_variant_time.py bwd both 2>&1 | grep -v amdgpu.ids | tee -a ${O}_ab.txt
done
for v in m16_abl m8_abl m8; do
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_$v.so timeout 120 python tools/dbg/strip_variant_time.py fwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}_ab.txt
done
echo "== 8-frame chunks: parity of the strip tests on that build"
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_m8.so timeout 600 python -m pytest tests/test_strip_gpu.py -m gpu -q > ${O}_m8_tests.log 2>&1; tail -n 6 ${O}_m8_tests.log | cut -c1-300
echo "== 8-frame chunks: kernel trace + PMC of the metric run"
NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_m8.so rocprofv3 --kernel-trace --stats -d ${O}_m8_trace -o run -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > ${O}_m8_trace.log 2>&1
sum m8_trace; grep -i "strip_kernel" ${O}_m8_trace.txt | head -3
for c in FETCH_SIZE WRITE_SIZE; do
  NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_m8.so rocprofv3 --kernel-trace --pmc $c -d ${O}_m8_pmc_$c -o run -- python bench.py --no-cpu-baseline --no-secondary --regions 0 --steps 5 --warmup 1 > ${O}_m8_pmc_$c.log 2>&1
  sum m8_pmc_$c --pmc; grep -i "strip_kernel" ${O}_m8_pmc_$c.txt | head -2
done
echo "== backward traffic of the new default (float64 / float32)"
for d in f64 f32; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d ${O}_bwd_${d}_pmc_$c -o run -- python tools/dbg/strip_variant_time.py bwd $d > ${O}_bwd_${d}_pmc_$c.log 2>&1
  sum bwd_${d}_pmc_$c --pmc; grep -i "strip_kernel" ${O}_bwd_${d}_pmc_$c.txt | head -2
done; done
echo "== the bench line"
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > ${O}_bench.log 2>&1
grep "^{" ${O}_bench.log | tail -1 > ${O}_bench.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r5_strip_bench.json"))
print("value %.4g  ms/step %.4f  roofline %s" % (r["value"], r["ms_per_step"], {k: r["roofline"].get(k) for k in ("frac", "peak_measured", "frac_of_measured", "kernel_ms", "kernel_ms_steady", "traffic")}))
for k, v in (r.get("secondary") or {}).items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "frac", "ms_hip_graph_replay", "ms_forward", "ms_backward", "error", "algo")})
PY
