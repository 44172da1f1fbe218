#!/bin/bash
# GPU call C: graphstep2 iteration -- headline parity, phase clocks, bench.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/c
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -q -x 2>&1 | tail -15 ) > $O/gpu_parity.log
( timeout 200 python tools/g2_phase_clocks.py ) > $O/g2_clocks.txt 2>&1
( timeout 200 python bench.py --no-cpu-baseline --rmse-links 0 ) > $O/bench.json 2> $O/bench.err
tail -3 $O/gpu_parity.log
cat $O/g2_clocks.txt | tail -45
python - $O/bench.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'gs avg_us', r.get('avg_us'), 'eager', r.get('avg_us_eager_events'), 'frac', r.get('frac'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
