#!/bin/bash
# the driver's form (20 steps after 5) with one graph launch of 20 steps against two of 10 / four of 5
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-groupab}; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
  for g in 10 5; do
    timeout 300 python bench.py --steps 20 --warmup 5 --group $g --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 --no-secondary > $O/g${g}_$i.json 2> $O/g${g}_$i.err
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/g*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step', d['config']['steps_per_graph_launch'], d['timing_check'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
