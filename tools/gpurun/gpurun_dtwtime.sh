#!/usr/bin/env bash
# DTW phase timers: rebuild dtw_fast with -DMLPG_DTW_TIMING on the GPU box (hipcc is there), run, restore nothing (box is scratch)
cd "$GRAFT_REPO_ROOT"
MLPG_HIP_EXTRA_FLAGS="-DMLPG_DTW_TIMING" python -c "
import sys; sys.path.insert(0,'.')
from nnmnkwii_amd.csrc import build; build.build(force=True)" > /dev/null 2>&1
python tools/dbg/dbg_dtw_timing.py 2>&1 | tail -2
