#!/bin/bash
# same-box A/B of the product library against variants over several configurations:
#   gpurun -- 'bash tools/gpu_ab_cfgs.sh <tag> "<variants>" "<configs>" <reps> [pytest args...]'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-abcfgs}; mkdir -p $O
VARS=$2; CFGS=$3; REPS=${4:-2}; shift 4
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for c in $CFGS; do
    for v in new $VARS; do
      lib=""; [ $v != new ] && lib=$PWD/igmc_amd/lib/libigmc_hip_$v.so
      env IGMC_LIB_PATH=$lib timeout 200 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/bench_${c}_${v}_$rep.json 2> $O/bench_${c}_${v}_$rep.err
    done
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
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get('kernels_us') or {}
        print('%-30s %7.0f subgraphs/s %7.2f us/step  fwd %.1f bwd %.1f  final loss %.9f' % (f.split('/')[-1][6:-5], d['value'], d['ms_per_step']*1e3, k.get('k_dl_fwd') or 0, k.get('k_dl_bwd') or 0, d.get('final_loss')))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
