#!/bin/bash
# GPU call F: full GPU suite (static cache, transfer eval, lean dropout, douban on the subgraph kernel, DP structures),
# bench lines: ml_1m driver-style (dp_structure leg), douban (now on k_graph_step2), ml_100k.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/f
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $O/gpu_tests.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > $O/smoke.log
( timeout 400 python bench.py --steps 20 --warmup 5 ) > $O/bench_ml1m_driver.json 2> $O/bench_ml1m_driver.err
( timeout 400 python bench.py --config douban ) > $O/bench_douban.json 2> $O/bench_douban.err
( timeout 300 python bench.py --config ml_100k --no-cpu-baseline ) > $O/bench_ml100k.json 2> $O/bench_ml100k.err
tail -8 $O/gpu_tests.log; cat $O/smoke.log
for f in $O/bench_ml1m_driver.json $O/bench_douban.json $O/bench_ml100k.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print(sys.argv[1].split('/')[-1], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), r.get('kernel'), 'avg_us', r.get('avg_us'), 'frac', r.get('frac'), 'traffic', r.get('traffic'), d['kernels_us'])
    print('   cpu', (d.get('cpu_baseline') or {}).get('value'), 'rmse', d.get('rmse'))
    print('   dp', d.get('dp_structure'), 'extraction', d.get('extraction'))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
