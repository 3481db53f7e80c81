#!/usr/bin/env bash
# Round 6, first GPU minutes: state of the head (-m gpu suite), what one small host-memory call costs piece by piece
# (tools/dbg/small_call_latency.hip), the literal drop-in calls as they are (bench_paths --only lit), where the Python time goes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_first
timeout 1500 python -m pytest tests -m gpu -q -x > ${O}_tests.log 2>&1
echo "pytest rc=$?" >> ${O}_tests.log
tail -4 ${O}_tests.log
echo "== small call latency"
timeout 300 tools/dbg/bin/small_call_latency > ${O}_small_call_latency.txt 2>&1; echo "rc=$?"
cat ${O}_small_call_latency.txt
echo "== literal calls (before)"
timeout 600 python tools/bench_paths.py --only lit > ${O}_lit_before.jsonl 2>${O}_lit_before.err; echo "rc=$?"
cat ${O}_lit_before.jsonl; tail -5 ${O}_lit_before.err
echo "== python profile"
timeout 300 python tools/dbg/lit_profile.py > ${O}_lit_profile.txt 2>&1; echo "rc=$?"
head -60 ${O}_lit_profile.txt
