#!/bin/bash
# per-step cost of short runs by launch structure: --group M forces pairs of groups of M; without it a run of <= 32 steps is ONE
# single-group launch (round 6)
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
for rep in 1 2 3 4; do
for cfg in "20 10" "20 0" "30 0" "40 20" "200 0"; do
  set -- $cfg
  g=""; [ $2 != 0 ] && g="--group $2"
  python bench.py --steps $1 --warmup 5 $g --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor --profile-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 group ${2}: %.2f us/step  (wall %.1f us; launch = %s steps)' % (d['ms_per_step']*1e3, d['timing_check']['wall_ms']*1e3, d['config']['steps_per_graph_launch']))"
done; done
