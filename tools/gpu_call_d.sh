#!/bin/bash
# GPU call D: extraction branch (balanced k_relm, lean arenas) + graphstep2 -- full GPU suite, bench, kernel trace.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/d
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/gpu_tests.log
( timeout 200 python bench.py --no-cpu-baseline --rmse-links 0 ) > $O/bench.json 2> $O/bench.err
( timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench20.json 2> $O/bench20.err
tail -3 $O/gpu_tests.log
for f in $O/bench.json $O/bench20.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'gs avg_us', r.get('avg_us'), 'eager', r.get('avg_us_eager_events'), 'frac', r.get('frac'), d['kernels_us'], d.get('rmse'))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
bash tools/prof_kernel_trace.sh
