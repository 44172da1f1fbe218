#!/bin/bash
# GPU call K: chained steps (Adam kernel writes the next step's images): bit-identity test, full suite, bench A/B, clocks.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/k
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/gpu_tests.log
tail -4 $O/gpu_tests.log
( timeout 300 python tools/g2_phase_clocks.py 2>&1 | grep -v "^B[123] \|^L[123] " | tail -16 ) > $O/phase_clocks.txt
cat $O/phase_clocks.txt
run() {  # name, env..., then -- args
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 $ARGS ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline'] or {}
    print('%-22s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r.get('avg_us'), 'frac %.3f'%r.get('frac'), d['kernels_us'])
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
ARGS=""
run base A=1
run selfseq IGMC_G2_SELF_SEQ=1
run base2 A=1
ARGS="--steps 20 --warmup 5"
run driver A=1
run driver_selfseq IGMC_G2_SELF_SEQ=1
ARGS="--steps 400 --warmup 20"
run base400 A=1
