#!/bin/bash
# same-box A/B of library variants built by tools/build_patch_variants.py:
#   gpurun -- 'bash tools/gpu_ab_variants.sh <tag> "prologue transpose both" [config] [reps]'
# interleaved runs of the product library ("base") and igmc_amd/lib/libigmc_hip_<name>.so, the headline bench line of each
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-variants}; mkdir -p $O
VARS=${2:-}; CFG=${3:-ml_1m}; REPS=${4:-2}
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in base $VARS; do
    lib=""; [ $v != base ] && lib=$PWD/igmc_amd/lib/libigmc_hip_$v.so
    env IGMC_LIB_PATH=$lib timeout 200 python bench.py --config $CFG --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  done
done
# the GPU suite's headline / parity files against every variant (TESTS=0 skips it)
if [ "${TESTS:-1}" != 0 ]; then
  for v in $VARS; do
    IGMC_LIB_PATH=$PWD/igmc_amd/lib/libigmc_hip_$v.so timeout 900 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_$v.log 2>&1
    echo "$v: $(tail -1 $O/pytest_$v.log)"
  done
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
