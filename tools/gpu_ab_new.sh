#!/bin/bash
# same-box A/B of the product library ("new") against older builds kept beside it:
#   gpurun -- 'bash tools/gpu_ab_new.sh <tag> "r4base ..." [config] [reps] [pytest files...]'
# interleaved bench lines (200 steps after 20) of every library, then the given GPU test files against the PRODUCT library
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-abnew}; mkdir -p $O
VARS=${2:-r4base}; CFG=${3:-ml_1m}; REPS=${4:-2}; shift 4
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in new $VARS; do
    lib=""; [ $v != new ] && lib=$PWD/igmc_amd/lib/libigmc_hip_$v.so
    env IGMC_LIB_PATH=$lib timeout 200 python bench.py --config $CFG --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  done
done
if [ $# -gt 0 ]; then
  timeout 1500 python -m pytest "$@" -m gpu -x -q > $O/pytest.log 2>&1
  echo "pytest: $(tail -1 $O/pytest.log)"
fi
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print('%-24s %7.0f subgraphs/s %7.2f us/step  dominant kernel %6.2f us  final loss %.9f' % (f.split('/')[-1][6:-5], d['value'], d['ms_per_step']*1e3, r.get('avg_us') or 0, d.get('final_loss')))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
