#!/usr/bin/env bash
# Round-5 soaks at the final head (logs -> profiles/r05_soak_*.log): every MLPG kernel family against the oracle / the natural-order
# kernel, the FIR form and its training step, fastdtw against the C oracle under both tie rules, the aligners.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 200 python tools/dbg/mlpg_algos_soak.py 90 2026 > gpurun_out/r05_soak_mlpg_algos.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soak_mlpg_algos.log; tail -n 3 gpurun_out/r05_soak_mlpg_algos.log
timeout 200 python tools/dbg/soak_strip.py 60 11 > gpurun_out/r05_soak_strip.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soak_strip.log; tail -n 2 gpurun_out/r05_soak_strip.log
timeout 200 python tools/dbg/fir_soak.py 60 99 > gpurun_out/r05_soak_fir.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soak_fir.log; tail -n 2 gpurun_out/r05_soak_fir.log
timeout 200 python tools/dbg/mlpg_soak.py 45 3 > gpurun_out/r05_soak_streams.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soak_streams.log; tail -n 2 gpurun_out/r05_soak_streams.log
timeout 200 python tools/dbg/dtw_soak.py 60 8 > gpurun_out/r05_soak_dtw.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soak_dtw.log; tail -n 2 gpurun_out/r05_soak_dtw.log
