#!/bin/bash
# GPU call G: lin1 coalescing check -- parity subset, phase clocks, bench lines.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/g
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/gpu_tests.log
( timeout 300 python tools/g2_phase_clocks.py 2>&1 ) > $O/phase_clocks.txt
( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 ) > $O/bench_ml1m_200.json 2> $O/bench_ml1m_200.err
( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dp-steps 0 ) > $O/bench_ml1m_driver.json 2> $O/bench_ml1m_driver.err
tail -3 $O/gpu_tests.log; cat $O/phase_clocks.txt
for f in $O/bench_ml1m_200.json $O/bench_ml1m_driver.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print(sys.argv[1].split('/')[-1], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), r.get('kernel'), 'avg_us', r.get('avg_us'), 'frac', r.get('frac'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
