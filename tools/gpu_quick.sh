#!/bin/bash
# quick kernel-iteration check: headline parity tests, phase clocks, two bench lines
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-quick}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 200 python tools/g2_phase_clocks.py > $O/phase_clocks.txt 2>&1; tail -47 $O/phase_clocks.txt
for i in 1 2; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_200_$i.json 2> $O/bench_200_$i.err; done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
