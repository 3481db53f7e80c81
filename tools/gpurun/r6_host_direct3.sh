#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/host_direct_ab3.txt
: > $O
python tools/dbg/host_sizes.py "shipped: direct >= 1200 KB when seen before, >= 4096 KB always; one-stream path up to 64 MB" big >> $O 2>&1
MLPG_HIP_HOST_DIRECT_KB=0 MLPG_HIP_HOST_SMALL_MB=6 python tools/dbg/host_sizes.py "staged, one-stream path up to 6 MB (the state before)" big >> $O 2>&1
grep -v amdgpu.ids $O
timeout 900 python -m pytest tests/test_literal_calls_gpu.py tests/test_host_multi_gpu.py -x -q > gpurun_out/direct_tests.log 2>&1
tail -n 4 gpurun_out/direct_tests.log
timeout 400 python tools/dbg/lit_soak.py 120 13 > gpurun_out/direct_soak.log 2>&1
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
