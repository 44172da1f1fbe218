#!/bin/bash
# basis-space finalize for R <= 32 (flixster) + experiment: 128 weight-gradient blocks per slice
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/z; mkdir -p $O; export PYTHONPATH=$ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -3 $O/gpu_tests.log
run() {
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 $ARGS ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print('%-16s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'rmse', (d.get('rmse') or {}).get('value'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1200:])
PY
}
ARGS="--config flixster"; run flixster A=1
ARGS="--config flixster"; run flixster_wg128 IGMC_LIB_PATH=$ROOT/igmc_amd/lib/exp/libigmc_hip_wg128.so
ARGS="--config ml_100k"; run ml100k A=1
ARGS="--config ml_100k"; run ml100k_wg128 IGMC_LIB_PATH=$ROOT/igmc_amd/lib/exp/libigmc_hip_wg128.so
ARGS="--config yahoo_music"; run yahoo A=1
