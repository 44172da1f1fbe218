#!/bin/bash
# The GPU suite, smoke() and the default bench line (all legs) on the current tree -- no profiling passes (kernel sources
# unchanged since the committed traffic record): the round's last session when only tests / host code moved
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-suite}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2), 'us/step frac', r.get('frac'), 'traffic', r.get('traffic'), 'sha', d.get('kernel_src_sha'))
        c=d.get('cpu_baseline') or {}
        print('   cpu', c.get('value'), (c.get('twin') or {}).get('value'), (c.get('extraction_twin') or {}).get('value'))
        for k,v in (d.get('secondary') or {}).items(): print('   ', k, v.get('value'), v.get('us_per_step'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
