#!/bin/bash
# how the next group's extraction launches are held back beside the steps (IGMC_EXTRACT_PACED: 2 = gate kernels polling the
# step counter, 1 = edges out of the step chain, 0 = not at all; MODES="mode:gate delay in us ..."), same box, interleaved, in the driver's 20-step form and the
# 200-step form:
#   gpurun -- 'bash tools/gpu_pacing_ab.sh <tag> [reps] [config] [pytest args...]'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-pacing}; mkdir -p $O
REPS=${2:-3}; CFG=${3:-ml_1m}; shift 3
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for mode in ${MODES:-2:10 2:0 2:5 2:15 1:0 0:0}; do
    for form in "20 5" "200 20"; do
      set -- $form "$@"; K=$1; W=$2; shift 2
      env IGMC_EXTRACT_PACED=${mode%%:*} IGMC_GATE_DELAY_US=${mode##*:} timeout 200 python bench.py --config $CFG --steps $K --warmup $W --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor --profile-steps 0 > $O/bench_m${mode}_k${K}_$rep.json 2> $O/bench_m${mode}_k${K}_$rep.err
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
        d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('timing_check') or {}
        print('%-16s %7.0f subgraphs/s %7.2f us/step  gpu %.2f us/step  host enqueue %.0f us' % (f.split('/')[-1][6:-5], d['value'], d['ms_per_step']*1e3, (t.get('gpu_event_ms') or 0)*1e3/d['steps'], (t.get('host_enqueue_ms') or 0)*1e3))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
PY
