#!/bin/bash
# GPU call B (round 2): the matrix-core subgraph kernel (graphstep2.hip) -- parity first, then speed A/B.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/b
mkdir -p $O
export PYTHONPATH=$ROOT
# 1. parity at the headline shape with the new kernel (fail fast), then the whole suite
( timeout 600 python -m pytest tests/test_gpu_headline.py -q -x 2>&1 | tail -30 ) > $O/gpu_headline.log
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/gpu_tests.log
# 2. speed: new vs old subgraph kernel, same build
for v in 2 1; do
  ( IGMC_GS_VERSION=$v timeout 200 python bench.py --no-cpu-baseline --rmse-links 0 ) > $O/bench_v$v.json 2> $O/bench_v$v.err
done
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err
# 3. first-replay cost of a captured graph
( timeout 200 python tools/exp_graph_launch.py ) > $O/exp_graph_launch.txt 2>&1
tail -4 $O/gpu_headline.log; tail -4 $O/gpu_tests.log
for f in $O/bench_v2.json $O/bench_v1.json $O/bench_driver.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d['roofline'] or {}
    print(sys.argv[1].split('/')[-1], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'gs avg_us', r.get('avg_us'), 'eager', r.get('avg_us_eager_events'), 'frac', r.get('frac'), d['kernels_us'])
    print('   cpu', d.get('cpu_baseline'))
except Exception as e:
    print(sys.argv[1], 'ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
cat $O/exp_graph_launch.txt | tail -8
