#!/usr/bin/env bash
# per-kernel durations of the FIR path (rocprofv3 kernel trace) at the config-3 and config-2 shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "64 500 60" "256 1000 60"; do
  tag=$(echo $shape | tr ' ' x)
  rocprofv3 --kernel-trace --stats -d gpurun_out/fir_$tag -o t -- python tools/dbg/fir_run.py $shape > /dev/null 2>&1
  echo "== $shape"; python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/fir_$tag/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v); print("%-70s n=%3d median %.2f us" % (k, len(v), v[len(v)//2] / 1e3))
PY
done
