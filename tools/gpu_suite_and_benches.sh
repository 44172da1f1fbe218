#!/bin/bash
# head role of k_head_train with its operands requested up front: GPU suite + bench lines of the per-layer paths
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/v; mkdir -p $O; export PYTHONPATH=$ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -3 $O/gpu_tests.log
run() {
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 $ARGS ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print('%-16s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'rmse', (d.get('rmse') or {}).get('value'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
}
ARGS="--config ml_100k"; run ml100k A=1
ARGS="--config flixster"; run flixster A=1
ARGS=""; run ml1m_a A=1
ARGS="--config ml_100k"; run ml100k_b A=1
ARGS=""; run ml1m A=1
