#!/bin/bash
# the driver's 20-step form under structure knobs of the step graph (same box, interleaved):
#   gpurun -- 'bash tools/gpu_driver_form.sh <tag> [reps]'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-driver_form}; mkdir -p $O
REPS=${2:-3}
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in default unpaced chunk1 chunk4 group5 nooverlap; do
    case $v in
      default) E="";;
      unpaced) E="IGMC_EXTRACT_PACED=0";;
      chunk1) E="IGMC_GROUP_EXTRACT_CHUNK=1";;
      chunk4) E="IGMC_GROUP_EXTRACT_CHUNK=4";;
      group5) E="X=1"; G="--group 5";;
      nooverlap) E="IGMC_NO_OVERLAP=1";;
    esac
    [ $v != group5 ] && G=""
    env $E timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor --profile-steps 0 $G > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
  done
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('timing_check') or {}
        print('%-22s %7.0f subgraphs/s %7.2f us/step  gpu %.1f us/step' % (f.split('/')[-1][6:-5], d['value'], d['ms_per_step']*1e3, (t.get('gpu_event_ms') or 0)*1e3/d['steps']))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
