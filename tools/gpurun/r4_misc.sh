#!/usr/bin/env bash
# round-4 odds and ends: autograd fallbacks, dtw flag words, host entry points, strip block dealing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_autograd_gpu.py tests/test_dtw_gpu.py tests/test_host_multi_gpu.py tests/test_parity_r2_gpu.py tests/test_strip_gpu.py tests/test_align_gpu.py -m gpu -x -q 2>&1 | tail -5
MLPG_HIP_HOST_TRACE=1 timeout 300 python tools/dbg/host_path_time.py 2>&1 | grep -v amdgpu.ids | awk 'NR%4==0 || /median/'
