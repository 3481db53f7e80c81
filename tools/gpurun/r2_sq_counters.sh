#!/usr/bin/env bash
# A few SQ counters of the strip kernel on the bench workload (separate passes, kernel trace only), for profiles/r02_notes.md
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=sq_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  f=$(find gpurun_out/$tag -name "*.db" | head -1)
  [ -n "$f" ] && python tools/rocpd_summary.py "$f" --pmc 2>/dev/null | grep "strip_kernel" | grep -E "SQ_" | awk '{print $(NF-2), $(NF-1), $NF}'
  rm -rf gpurun_out/$tag gpurun_out/$tag.log
done
