#!/bin/bash
# GPU call M2: is the host thread throttled (cgroup CPU quota) during the enqueue of the timed steps?
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/m
mkdir -p $O
export PYTHONPATH=$ROOT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
EXTRA=""
for i in 1 2 3 4 5 6 7 8; do
  a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  ( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 $EXTRA ) > $O/b$i.json 2> $O/b$i.err
  b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  python - $O/b$i.json "$a" "$b" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), {k: round(v,1) for k,v in d['timing_check'].items()}, 'nr_throttled', sys.argv[2], '->', sys.argv[3])
PY
done
echo "--- driver-style x6"
EXTRA="--steps 20 --warmup 5"
for i in 11 12 13 14 15 16; do
  a=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  ( timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 $EXTRA ) > $O/b$i.json 2> $O/b$i.err
  b=$(grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}')
  python - $O/b$i.json "$a" "$b" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), {k: round(v,1) for k,v in d['timing_check'].items()}, 'nr_throttled', sys.argv[2], '->', sys.argv[3])
PY
done
( timeout 400 python bench.py --steps 20 --warmup 5 --dp-steps 0 ) > $O/full.json 2> $O/full.err
python - $O/full.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(round(d['value']), d['cpu_baseline'])
PY
