#!/bin/bash
# the -m gpu suite + smoke + one short bench line per config (after a change that should not move any number)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-suite}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for c in ml_1m ml_100k flixster; do
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_$c.json 2> $O/bench_$c.err
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step')
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
