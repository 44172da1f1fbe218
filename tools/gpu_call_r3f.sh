#!/bin/bash
# round 3, GPU session F: group extraction (one launch per stage for a whole group) vs arena-by-arena extraction, same box
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r3f}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
timeout 600 python tools/exp_repro.py 5 > $O/repro.txt 2>&1; echo "repro rc=$?"; grep "==\|DIFF\|RAISED\|Error" $O/repro.txt | head
for i in 1 2; do
  IGMC_NO_GROUP_EXTRACT=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/perarena_$i.json 2> $O/perarena_$i.err
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/group_$i.json 2> $O/group_$i.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/group_prof.json 2> $O/group_prof.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/group_driver.json 2> $O/group_driver.err
timeout 300 python bench.py --config douban --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/group_douban.json 2> $O/group_douban.err
timeout 300 python bench.py --config ml_100k --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/group_ml100k.json 2> $O/group_ml100k.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
