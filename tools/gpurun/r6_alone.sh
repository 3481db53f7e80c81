#!/usr/bin/env bash
# Round 6: what does an item cost a workgroup that has its CU to itself?  (grid capped at 256 = one workgroup per CU) + PMC passes of the shipped kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_alone
: > ${O}.txt
for cap in 0 256 128; do
echo "== grid cap $cap" | tee -a ${O}.txt
MLPG_STRIP_GRID_CAP=$cap timeout 120 python tools/dbg/strip_variant_time.py fwd f64 2>&1 | grep -v amdgpu.ids | tee -a ${O}.txt
MLPG_STRIP_GRID_CAP=$cap NNMNKWII_AMD_SO=$PWD/tools/dbg/bin/libmlpg_hip_trace.so timeout 120 python tools/dbg/strip_trace.py 2>&1 | grep -v "amdgpu.ids\|XCD [1-7]:\|loading per" | tee -a ${O}.txt
done
# PMC passes of the shipped kernel (bench metric only), one counter set per pass
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "VALUBusy" "MemUnitStalled" "MemUnitBusy" "OccupancyPercent" "LDSBankConflict" "VALUUtilization" "FetchSize WriteSize" "L2CacheHit" "TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_')
  d=gpurun_out/pmc_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $d -o run -- python bench.py --no-cpu-baseline --no-secondary --no-traffic --regions 0 --steps 5 --warmup 1 --precondition 0 > ${O}_pmc_$tag.log 2>&1
  f=$(find $d -name "*.db" | head -1)
  if [ -n "$f" ]; then python - "$f" "$set" <<'PY' | tee -a ${O}.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
try:
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%strip_kernel%' group by kernel_name, counter_name").fetchall()
    for k, c, v, n in rows:
        print("PMC %-34s %16.1f  (mean of %d dispatches of %s)" % (c, v, n, k.split("(")[0][-60:]))
    if not rows: print("PMC", sys.argv[2], ": no rows")
except Exception as e:
    print("PMC", sys.argv[2], "failed:", e)
PY
  else echo "PMC $set: no database (rc)" | tee -a ${O}.txt; tail -3 ${O}_pmc_$tag.log; fi
  rm -rf $d
done
