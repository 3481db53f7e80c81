#!/usr/bin/env bash
# round 6: longer soaks of every family at the head (different seeds than r6_final.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift; timeout $1 "${@:2}" > gpurun_out/r06_soakl_$name.log 2>&1; echo "rc=$?" >> gpurun_out/r06_soakl_$name.log; tail -n 3 gpurun_out/r06_soakl_$name.log | cut -c1-400; }
run lit 460 python tools/dbg/lit_soak.py 400 911
run strip 700 python tools/dbg/soak_strip.py 600 912
run mlpg_algos 400 python tools/dbg/mlpg_algos_soak.py 300 913
run streams 300 python tools/dbg/mlpg_soak.py 200 914
run fir 200 python tools/dbg/fir_soak.py 120 915
run dtw 300 python tools/dbg/dtw_soak.py 200 916
run align 200 python tools/dbg/align_soak.py 100 917
run lit_threads 400 python tools/dbg/lit_threads_soak.py 300 6 918
