#!/usr/bin/env bash
# round 6: instruction-cache counters of the headline kernel (50 KB of straight-line code in a 64 KB cache shared by two CUs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
sum() { f=$(find gpurun_out/$1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" $2 > gpurun_out/$1.txt 2>&1; rm -rf gpurun_out/$1; }
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"; do
  tag=r06_pmc_icache_$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/$tag -o run -- python bench.py --no-cpu-baseline --no-secondary --no-traffic --regions 0 --precondition 0 --steps 5 --warmup 1 > gpurun_out/$tag.log 2>&1
  sum $tag --pmc
  echo "== $set"; grep -i "strip_kernel" gpurun_out/$tag.txt | head -2 | cut -c1-260; tail -n 2 gpurun_out/$tag.log | cut -c1-200
done
