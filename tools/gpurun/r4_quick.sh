#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_const_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_paths.py --only c2g 2>&1 | grep -v amdgpu | cut -c1-200
