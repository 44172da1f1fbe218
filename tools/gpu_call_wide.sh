#!/bin/bash
# relation groups on the dense layers: the new flixster tests first, then the whole suite + bench lines
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-wide}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "flixster" > $O/pytest_flixster.log 2>&1; echo "flixster rc=$?"; tail -5 $O/pytest_flixster.log
bash tools/gpu_suite.sh ${1:-wide}
for c in flixster; do
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/full_$c.json 2> $O/full_$c.err
  python - $O/full_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step']*1e3,1), d.get('kernels_us'))
PY
done
