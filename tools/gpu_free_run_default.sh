#!/bin/bash
# free-running prefetch as the default: GPU suite, bench lines (driver form + 200 steps), dispatch timeline
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/fr2; mkdir -p $O; export PYTHONPATH=$ROOT
timeout 600 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -4 $O/gpu_tests.log
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench_ml1m_driver_nocpu.json 2> $O/bench_ml1m_driver_nocpu.err
( timeout 200 python bench.py --no-cpu-baseline ) > $O/bench_ml1m_200.json 2> $O/bench_ml1m_200.err
for f in $O/bench_ml1m_driver_nocpu.json $O/bench_ml1m_200.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print(sys.argv[1].split('/')[-1], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r['avg_us'], 'frac %.3f'%r['frac'], 'traffic', r['traffic'], 'rmse', (d.get('rmse') or {}).get('value'), 'dp', d.get('dp_structure_us'))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
bash tools/gpu_timeline.sh > $O/tl.log 2>&1; tail -2 $O/tl.log
