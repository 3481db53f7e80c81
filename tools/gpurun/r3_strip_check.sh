#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_strip_gpu.py tests/test_mlpg_gpu.py tests/test_parity_r2_gpu.py -x -q 2>&1 | tail -5
timeout 200 python tools/dbg/pipe_time.py 2>&1 | grep -v amdgpu
