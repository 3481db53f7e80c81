#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_literal_calls_gpu.py -x -q 2>&1 | tail -n 3
python tools/bench_paths.py --only lit > gpurun_out/lit3.jsonl 2>gpurun_out/lit3.err
grep -o '"path": "lit-unit[^}]*' gpurun_out/lit3.jsonl | cut -c1-400
tail -n 3 gpurun_out/lit3.err
