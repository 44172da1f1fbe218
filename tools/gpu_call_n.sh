#!/bin/bash
# GPU call N: item-side transposition with batched LDS reads -- parity subset, phase clocks, bench x3.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/n
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 ) > $O/gpu_tests.log
tail -2 $O/gpu_tests.log
( timeout 300 python tools/g2_phase_clocks.py 2>&1 | grep -v "^B[123] \|^L[23] " | tail -22 ) > $O/phase_clocks.txt
cat $O/phase_clocks.txt
for i in 1 2 3; do
  ( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 ) > $O/b$i.json 2> $O/b$i.err
  python - $O/b$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r['avg_us'], 'frac %.3f'%r['frac'], 'traffic', r['traffic'], {k: round(v,1) for k,v in d['timing_check'].items()})
PY
done
