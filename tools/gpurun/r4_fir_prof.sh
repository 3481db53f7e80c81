#!/usr/bin/env bash
# per-kernel durations of the FIR path (rocprofv3 kernel trace) at the config-3 and config-2 shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in "64 500 60" "256 1000 60"; do
  tag=$(echo $shape | tr ' ' x)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fir_$tag -o t -- python tools/dbg/fir_run.py $shape > /dev/null 2>&1
  echo "== $shape"
  f=$(find gpurun_out/fir_$tag -name 't_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-150 | head -12
done
