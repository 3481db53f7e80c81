#!/usr/bin/env bash
# Round-5 soaks at the final head, longer (logs -> profiles/r05_soakf_*.log); the strip soak now draws failing pivots as well.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 260 python tools/dbg/soak_strip.py 150 511 > gpurun_out/r05_soakf_strip.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_strip.log; tail -n 3 gpurun_out/r05_soakf_strip.log
timeout 260 python tools/dbg/mlpg_algos_soak.py 150 512 > gpurun_out/r05_soakf_mlpg_algos.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_mlpg_algos.log; tail -n 3 gpurun_out/r05_soakf_mlpg_algos.log
timeout 200 python tools/dbg/fir_soak.py 60 513 > gpurun_out/r05_soakf_fir.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_fir.log; tail -n 2 gpurun_out/r05_soakf_fir.log
timeout 200 python tools/dbg/mlpg_soak.py 60 514 > gpurun_out/r05_soakf_streams.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_streams.log; tail -n 2 gpurun_out/r05_soakf_streams.log
timeout 200 python tools/dbg/dtw_soak.py 90 515 > gpurun_out/r05_soakf_dtw.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_dtw.log; tail -n 2 gpurun_out/r05_soakf_dtw.log
timeout 200 python tools/dbg/align_soak.py 40 516 > gpurun_out/r05_soakf_align.log 2>&1; echo "rc=$?" >> gpurun_out/r05_soakf_align.log; tail -n 2 gpurun_out/r05_soakf_align.log
