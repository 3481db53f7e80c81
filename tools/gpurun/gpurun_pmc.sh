#!/usr/bin/env bash
# PMC passes over the bench command (own runs, kernel-trace only). usage: gpurun_pmc.sh <tag> "<counters pass1>" ["<pass2>" ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -f gpurun_out/counters_list.txt ] || rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass -d gpurun_out/${tag}_p$i -o run -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${tag}_p$i.log 2>&1
  tail -1 gpurun_out/${tag}_p$i.log | cut -c1-200
done
ls gpurun_out/${tag}_p*/
