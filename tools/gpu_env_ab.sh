#!/bin/bash
# same-box A/B of environment knobs in the driver's 20-step form and the 200-step form:
#   gpurun -- 'bash tools/gpu_env_ab.sh <tag> <reps> "name1:VAR=val,VAR2=val name2:..." [forms: "20:5 200:20"]'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-envab}; mkdir -p $O
REPS=${2:-3}; VARS=${3:-base:X=1}; FORMS=${4:-20:5 200:20}
export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in $VARS; do
    name=${v%%:*}; envs=$(echo "${v#*:}" | tr ',' ' ')
    for form in $FORMS; do
      K=${form%%:*}; W=${form##*:}
      env $envs timeout 200 python bench.py --steps $K --warmup $W --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor --profile-steps 0 > $O/bench_${name}_k${K}_$rep.json 2> $O/bench_${name}_k${K}_$rep.err
    done
  done
done
python - "$O" <<'PY'
import json,glob,sys,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); t=d.get('timing_check') or {}
        key=f.split('/')[-1][6:-5].rsplit('_',1)[0]
        acc[key].append(d['ms_per_step']*1e3)
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-300:])
for k,v in sorted(acc.items()):
    print('%-28s us/step: %s   median %.2f' % (k, ' '.join('%.2f'%x for x in v), sorted(v)[len(v)//2]))
PY
