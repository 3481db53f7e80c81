#!/usr/bin/env bash
# host-memory entry points over a device list + output pre-fault: tests and wall clock
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_host_multi_gpu.py tests/test_parity_r2_gpu.py tests/test_dtw_gpu.py -m gpu -x -q 2>&1 | tail -5
MLPG_HIP_HOST_TRACE=1 timeout 300 python tools/dbg/host_path_time.py 2>&1 | grep -v amdgpu.ids | awk 'NR%4==0 || /median/'
