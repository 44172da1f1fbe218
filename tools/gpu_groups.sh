#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-groups}; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for st in 200 400 256; do
  timeout 300 python bench.py --steps $st --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/s${st}_$rep.json 2> $O/s${st}_$rep.err
done; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step', d['config']['steps_per_graph_launch'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
