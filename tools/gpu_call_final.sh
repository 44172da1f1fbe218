#!/bin/bash
# Round-end GPU session: full GPU suite, smoke, bench lines (driver-style with CPU baseline, 200 steps, douban, ml_100k),
# rocprofv3 kernel trace + PMC passes of the headline configuration.   IGMC_COMMIT=<hash> bash tools/gpu_call_final.sh
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/final
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/gpu_tests.log
( timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 ) > $O/smoke.log
( timeout 400 python bench.py --steps 20 --warmup 5 ) > $O/bench_ml1m_driver.json 2> $O/bench_ml1m_driver.err
( timeout 300 python bench.py --no-cpu-baseline ) > $O/bench_ml1m_200.json 2> $O/bench_ml1m_200.err
( timeout 400 python bench.py --config douban ) > $O/bench_douban.json 2> $O/bench_douban.err
( timeout 400 python bench.py --config ml_100k ) > $O/bench_ml100k.json 2> $O/bench_ml100k.err
bash tools/profile_round.sh ml_1m
tail -3 $O/gpu_tests.log; cat $O/smoke.log
for f in $O/bench_ml1m_driver.json $O/bench_ml1m_200.json $O/bench_douban.json $O/bench_ml100k.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print(sys.argv[1].split('/')[-1], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), r.get('kernel'), 'avg_us', r.get('avg_us'), 'frac', r.get('frac'), 'traffic', r.get('traffic'), d['kernels_us'])
    print('   cpu', (d.get('cpu_baseline') or {}).get('value'), 'rmse', (d.get('rmse') or {}).get('value'), 'dp', d.get('dp_structure_us'), 'ext', (d.get('extraction') or {}).get('us_per_step'))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
head -14 gpurun_out/prof/kernel_stats.txt; cat gpurun_out/prof/pmc_traffic.json
