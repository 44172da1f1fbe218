#!/bin/bash
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/y; mkdir -p $O; export PYTHONPATH=$ROOT
run() {
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --rmse-links 0 ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print('%-16s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r['avg_us'], 'ext', (d.get('extraction') or {}).get('us_per_step'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run base A=1
run nosplit IGMC_EXTRACT_SPLIT=0
run nooverlap IGMC_NO_OVERLAP=1
