#!/usr/bin/env bash
# the whole -m gpu suite and the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_full_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_full_tests.log
tail -5 gpurun_out/r4_full_tests.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r4_full_bench.log 2>&1
echo "bench rc=$?"
grep "^{" gpurun_out/r4_full_bench.log | tail -1 > gpurun_out/r4_full_bench.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4_full_bench.json"))
print("value %.4g  ms/step %.4f  roofline %s" % (r["value"], r["ms_per_step"], {k: r["roofline"].get(k) for k in ("frac", "peak_measured", "frac_of_measured", "kernel_ms", "kernel_ms_steady")}))
for k, v in (r.get("secondary") or {}).items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "frac", "frac_of_measured", "pairs_per_s", "frames_per_s", "ms_hip_graph_replay", "ms_forward", "ms_backward", "error", "algo")})
PY
tail -3 gpurun_out/r4_full_bench.log | cut -c1-300
