#!/usr/bin/env bash
# Round 5: every secondary path, AUTO's routing table and the narrow-stream table with gpu_time()'s settle phase (clocks settled)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python tools/bench_paths.py > gpurun_out/r05_paths_settled.log 2>&1
grep '"path"' gpurun_out/r05_paths_settled.log > gpurun_out/r05_paths_settled.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r05_paths_settled.jsonl'):
    d = json.loads(l)
    print("%-60s %9.4f  %s" % (d['path'], d['ms'], ("%.3f" % d['roofline_frac']) if d.get('roofline_frac') else ""))
PY
timeout 600 python tools/auto_routing.py > gpurun_out/r5_auto_routing_settled.log 2>&1; tail -n 30 gpurun_out/r5_auto_routing_settled.log | cut -c1-200
timeout 200 python tools/dbg/narrow_time.py frame 2>&1 | grep -v amdgpu.ids | grep float64 > gpurun_out/r05_tr_narrow_settled.txt; head -8 gpurun_out/r05_tr_narrow_settled.txt
