#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-chunks}; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do for c in 0 2 4 8 16; do
  IGMC_GROUP_EXTRACT_CHUNK=$c timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/c${c}_$rep.json 2> $O/c${c}_$rep.err
done; done
for c in 0 4; do IGMC_GROUP_EXTRACT_CHUNK=$c timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/prof_c${c}.json 2> $O/prof_c${c}.err; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
