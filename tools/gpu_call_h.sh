#!/bin/bash
# GPU call H: extraction split + LDS padding A/B, driver-style group sizing; parity subset first.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/h
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 ) > $O/gpu_tests.log
tail -3 $O/gpu_tests.log
run() {  # name, env...
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print('%-22s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r.get('avg_us'), 'frac %.3f'%r.get('frac'), d['kernels_us'], 'ext', round(d['extraction']['us_per_step'],1))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
run default A=1
run nopad IGMC_EXTRACT_LDS_PAD=0
run nosplit IGMC_EXTRACT_SPLIT=0
run nopad_nosplit IGMC_EXTRACT_LDS_PAD=0 IGMC_EXTRACT_SPLIT=0
run pad32k IGMC_EXTRACT_LDS_PAD=32768
run default_again A=1
( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --dp-steps 0 ) > $O/bench_driver.json 2> $O/bench_driver.err
python - $O/bench_driver.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('driver-style', round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), r['avg_us'], r['frac'])
PY
