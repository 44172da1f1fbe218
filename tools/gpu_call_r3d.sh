#!/bin/bash
# round 3, GPU session D: full suite, driver-form bench with the primed graph, DGCNN_RS through the step graph, small datasets
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r3d}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -8 $O/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 > $O/bench_200.json 2> $O/bench_200.err
timeout 300 python bench.py --config douban --dgcnn-rs --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_dgcnn_douban.json 2> $O/bench_dgcnn_douban.err
timeout 300 python bench.py --config flixster --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_flixster.json 2> $O/bench_flixster.err
timeout 300 python bench.py --config yahoo_music --steps 64 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_yahoo.json 2> $O/bench_yahoo.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step', d['config'].get('steps_per_graph_launch'), 'frac', r.get('frac'), r.get('kernel'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
