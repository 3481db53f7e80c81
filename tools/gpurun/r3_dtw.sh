#!/usr/bin/env bash
# DTW kernel check: parity tests, kernel times, phase timers (timing variant prebuilt into tools/dbg/bin, see README)
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_dtw_gpu.py tests/test_align_gpu.py tests/test_parity_r2_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python tools/bench_paths.py --only c4 2>&1 | grep -v Warn | cut -c1-400
if [ -f tools/dbg/bin/libmlpg_hip_dtwtiming.so ]; then
  NNMNKWII_AMD_SO=tools/dbg/bin/libmlpg_hip_dtwtiming.so timeout 200 python tools/dbg/dbg_dtw_timing.py 2>&1 | tail -2
fi
