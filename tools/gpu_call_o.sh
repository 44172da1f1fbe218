#!/bin/bash
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/o; mkdir -p $O; export PYTHONPATH=$ROOT
( timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 ) > $O/b.json 2> $O/b.err
python - $O/b.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['value']), d['dp_structure']); print(d['extraction']); print(d['roofline']['traffic'], d['roofline']['frac'])
PY
