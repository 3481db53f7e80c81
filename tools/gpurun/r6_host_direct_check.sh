#!/bin/bash
# round 6: parity of the host paths after arrays >= 1.2 MB go straight from / to the caller's memory and the one-stream path takes calls up to 64 MB
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_literal_calls_gpu.py tests/test_host_multi_gpu.py tests/test_mlpg_gpu.py tests/test_autograd_gpu.py tests/test_align_gpu.py tests/test_streams_gpu.py -x -q > gpurun_out/direct_tests.log 2>&1
tail -n 6 gpurun_out/direct_tests.log
timeout 400 python tools/dbg/lit_soak.py 180 11 > gpurun_out/direct_soak.log 2>&1
tail -n 3 gpurun_out/direct_soak.log
python tools/bench_paths.py --only lit > gpurun_out/direct_lit.jsonl 2>gpurun_out/direct_lit.err
python - <<'PY'
import json
for l in open('gpurun_out/direct_lit.jsonl'):
    try: d = json.loads(l)
    except Exception: continue
    if str(d.get('path', '')).startswith('lit'):
        print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items() if k in ('path', 'us_per_call', 'us_per_call_min', 'cpu_us_per_call', 'ms', 'us_forward', 'us_forward_backward', 'us_paramgen_mlpg_grad')})
PY
