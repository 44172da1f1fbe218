#!/bin/bash
# data-parallel step check: the GPU suite, then the default bench (dp_structure leg) twice
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-dp}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --rmse-links 0 > $O/bench_$i.json 2> $O/bench_$i.err
done
timeout 300 python bench.py --config ml_100k --steps 200 --warmup 20 --no-cpu-baseline --rmse-links 0 > $O/bench_ml_100k.json 2> $O/bench_ml_100k.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step', 'dp_structure', d.get('dp_structure'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
