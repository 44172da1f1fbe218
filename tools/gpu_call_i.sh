#!/bin/bash
# GPU call I: full GPU suite after the extraction split; driver-style bench (primed graph groups); g2 without overlap.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/i
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 ) > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
run() {  # name, args...
  local name=$1; shift
  ( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 "$@" ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print('%-22s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r.get('avg_us'), 'frac %.3f'%r.get('frac'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run driver --steps 20 --warmup 5
run driver2 --steps 20 --warmup 5
run default
run nooverlap --no-overlap
run w9s40 --steps 40 --warmup 9
