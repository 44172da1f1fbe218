#!/bin/bash
# GPU call Q: dense per-layer kernels (k_dl_layer) -- cap-200 parity, full suite, bench ml_100k with / without them.
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/q; mkdir -p $O; export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "ml100k" 2>&1 | tail -12 ) > $O/t1.log; tail -12 $O/t1.log
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.log; tail -4 $O/gpu_tests.log
run() {
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 $ARGS ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print('%-16s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), r.get('kernel'), 'avg_us %.1f'%r.get('avg_us'), 'frac %.3f'%r.get('frac'), 'rmse', (d.get('rmse') or {}).get('value'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
}
ARGS="--config ml_100k"
run ml100k_dl A=1
run ml100k_dl2 A=1
ARGS=""
run ml1m A=1
