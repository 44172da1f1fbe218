#!/bin/bash
# bench lines of the dense-layer configurations (flixster, ml_10m_lite, ml_100k), the headline and DGCNN_RS on douban + the gpu suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-wide4}; mkdir -p $O
export TMPDIR=/tmp
for c in flixster ml_10m_lite ml_100k ml_1m; do
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/full_$c.json 2> $O/full_$c.err
done
timeout 300 python bench.py --dgcnn-rs --config douban --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/full_dgcnn.json 2> $O/full_dgcnn.err
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/full_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step']*1e3,1), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
