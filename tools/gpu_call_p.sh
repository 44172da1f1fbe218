#!/bin/bash
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/p; mkdir -p $O; export PYTHONPATH=$ROOT
for sr in 8 2 4 16 1 8; do
  ( IGMC_RELM_SLICES=$sr timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --rmse-links 0 ) > $O/b$sr.json 2> $O/b$sr.err
  python - $O/b$sr.json $sr <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print('slices', sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r['avg_us'], 'k_relm eager %.1f'%d['kernels_us']['k_relm'])
PY
done
