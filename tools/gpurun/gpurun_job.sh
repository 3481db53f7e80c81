#!/usr/bin/env bash
# helper used with gpurun: tests, bench, rocprof summary
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests -m gpu -q 2>&1 | grep -E "^(FAILED|ERROR|[0-9]+ (passed|failed))|assert |Error" | head -40
python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench.json
