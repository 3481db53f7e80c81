#!/usr/bin/env bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_autograd_gpu.py tests/test_parity_r2_gpu.py -x -q -k "fused or 64_pairs or config3" 2>&1 | tail -6
timeout 300 python tools/bench_paths.py --only c3 2>&1 | grep -v amdgpu | cut -c1-400
