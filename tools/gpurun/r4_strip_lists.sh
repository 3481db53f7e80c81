#!/usr/bin/env bash
# long utterances dealt to the per-XCD work lists in blocks: parity (strip + streams suites) and the c2k paths
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_strip_gpu.py tests/test_streams_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/bench_paths.py --only c2k 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print('%-60s %8.4f ms  frac %.3f' % (r['path'], r['ms'], r.get('roofline_frac') or 0))
"
