#!/usr/bin/env bash
# Round 6: the start ramp as shipped (default 20 us): strip suites + soak, and every strip path with the ramp off / on
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_ramp
timeout 1200 python -m pytest tests/test_strip_gpu.py tests/test_strip_tr_gpu.py tests/test_streams_gpu.py tests/test_mlpg_gpu.py tests/test_parity_r2_gpu.py tests/test_soak_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python tools/dbg/soak_strip.py 60 606 2>&1 | tail -3 | tee ${O}_soak.log
for us in 0 20 0 20; do
  echo "== ramp $us us"
  MLPG_STRIP_STAGGER_US=$us timeout 600 python tools/bench_paths.py --only c2k,c2b,c5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        if 'ms' in r and ('strip' in r['path'] or r['path'].startswith(('c2b','c2-','c5','long','c2t'))): print('  %-62s %.4f ms' % (r['path'], r['ms']))" | tee -a ${O}_paths_$us.txt
done
