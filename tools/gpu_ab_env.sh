#!/bin/bash
# same-box A/B of one environment switch:  bash tools/gpu_ab_env.sh <tag> VAR A_VALUE B_VALUE [config]
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-abenv}; mkdir -p $O
VAR=$2; A=$3; Bv=$4; CFG=${5:-ml_1m}
export TMPDIR=/tmp
for i in 1 2 3; do
  for v in A B; do
    val=$A; [ $v = B ] && val=$Bv
    env $VAR=$val timeout 300 python bench.py --config $CFG --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err
  done
done
python - "$O" "$VAR" "$A" "$Bv" <<'PY'
import json,glob,sys
O,var,a,b=sys.argv[1:5]
for v,val in (('A',a),('B',b)):
    xs=[]
    for f in sorted(glob.glob(O+'/bench_%s_*.json'%v)):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); xs.append((d['ms_per_step']*1e3, d['value'], d.get('final_loss')))
        except Exception as e:
            print(f,'ERR',e, open(f.replace('.json','.err')).read()[-800:])
    print('%s=%s'%(var,val), ' '.join('%.2f us (%.0f/s, loss %r)'%x for x in xs))
PY
