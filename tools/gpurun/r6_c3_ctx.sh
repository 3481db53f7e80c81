#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: d = json.loads(l)
    except Exception: continue
    if d.get('path') in ('c3-unit-variance-autograd-fwd+bwd', 'c3-fused-unit-mse-step'):
        print('   ', d['path'], {k: round(v, 4) for k, v in d.items() if k.startswith('ms') and isinstance(v, float)})
PY
}
for keys in c3 litq,c3 c2b,c3 c2g,c2t,c3; do
  python tools/bench_paths.py --only $keys > gpurun_out/ctx.jsonl 2>/dev/null
  echo "== bench_paths --only $keys"; show gpurun_out/ctx.jsonl
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ctx_bench.json 2>/dev/null
echo "== bench.py --no-cpu-baseline"
python - <<'PY'
import json
d = json.load(open('gpurun_out/ctx_bench.json'))
for k in ('c3-unit-variance-autograd-fwd+bwd', 'c3-fused-unit-mse-step'):
    print('   ', k, {a: round(b, 4) for a, b in d['secondary'][k].items() if a.startswith('ms') and isinstance(b, float)})
PY
