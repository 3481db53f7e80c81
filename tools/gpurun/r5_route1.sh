#!/usr/bin/env bash
# Round 5: the 3-strip level-3 window with its fallback ladder, shipped form: whole suite, strip soaks (incl. tight dynamic variances: the
# ladder's other rungs), timing of every strip path
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r5_route1
timeout 1200 python -m pytest tests -m gpu -q > ${O}_tests.log 2>&1; echo "pytest rc=$?" >> ${O}_tests.log; tail -n 3 ${O}_tests.log
timeout 300 python tools/dbg/soak_strip.py 120 31 > ${O}_soak_strip.log 2>&1; echo "rc=$?" >> ${O}_soak_strip.log; tail -n 2 ${O}_soak_strip.log
timeout 300 python tools/dbg/mlpg_algos_soak.py 60 77 > ${O}_soak_algos.log 2>&1; echo "rc=$?" >> ${O}_soak_algos.log; tail -n 2 ${O}_soak_algos.log
timeout 200 python tools/dbg/mlpg_soak.py 40 > ${O}_soak_streams.log 2>&1; echo "rc=$?" >> ${O}_soak_streams.log; tail -n 2 ${O}_soak_streams.log
timeout 120 python tools/dbg/strip_variant_time.py all both 2>&1 | grep -v amdgpu.ids | tee ${O}_times.txt
timeout 300 python tools/bench_paths.py --only c2k,c2t,c5q 2>&1 | grep '"path"' | tee ${O}_paths.jsonl | cut -c1-170
