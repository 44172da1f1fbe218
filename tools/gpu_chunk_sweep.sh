#!/bin/bash
# same-box sweep of the batches per extraction launch (IGMC_GROUP_EXTRACT_CHUNK) under paced extraction, headline configuration
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-chunks}; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for c in ${CHUNKS:-2 4 5 8}; do
    IGMC_GROUP_EXTRACT_CHUNK=$c timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 --no-secondary --no-floor > $O/bench_c${c}_$rep.json 2> $O/bench_c${c}_$rep.err
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_c*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], '%.0f subgraphs/s %.2f us/step' % (d['value'], d['ms_per_step']*1e3))
    except Exception as e:
        print(f, 'ERR', e)
PY
