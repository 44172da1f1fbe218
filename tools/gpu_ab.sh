#!/bin/bash
# same-box A/B of two builds of the library: IGMC_LIB_PATH switches bench.py between them
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-ab}; mkdir -p $O
export TMPDIR=/tmp
BASE=$PWD/igmc_amd/lib/libigmc_hip_base.so
for i in 1 2 3; do
  IGMC_LIB_PATH=$BASE timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 --no-secondary --no-floor > $O/base_$i.json 2> $O/base_$i.err
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 --no-secondary --no-floor > $O/new_$i.json 2> $O/new_$i.err
done
IGMC_LIB_PATH=$BASE timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/base_prof.json 2> $O/base_prof.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/new_prof.json 2> $O/new_prof.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
