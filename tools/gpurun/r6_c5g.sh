#!/usr/bin/env bash
# Round 6: forward_streams with global variances no longer merges streams whose narrow members take the transposed form; the short first staging piece
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_c5g
: > ${O}.txt
for r in 1 2; do
timeout 600 python tools/bench_paths.py --only c5q,litq 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l)
        v = ('%.4f ms' % r['ms']) if 'ms' in r else ('%.1f us (cpu %.1f)' % (r.get('us_per_call', r.get('us_forward', 0)), r.get('cpu_us_per_call', r.get('cpu_us_forward', 0))))
        print('  %-62s %s %s' % (r['path'], v, r.get('launches','')))" | tee -a ${O}.txt
done
