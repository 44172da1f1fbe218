#!/bin/bash
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/r; mkdir -p $O; export PYTHONPATH=$ROOT
( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --config ml_100k --no-overlap ) > $O/b.json 2> $O/b.err
python - $O/b.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), d['kernels_us'])
PY
