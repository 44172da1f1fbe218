#!/bin/bash
# round 6, first session: the data-parallel path (self-spawned ranks on ONE GPU: dry runs; the two-rank tests; the
# two-device test skips here) + the default bench line of the round's starting build as the same-round baseline
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-r6a}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "two_ranks or captured_all_reduce or step_gate" > $O/pytest_dp.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_dp.log; tail -4 $O/pytest_dp.log
for n in 2 4; do
  [ $n -gt 2 ] && export IGMC_GRAPH_STEP=0
  IGMC_DIST_BACKEND=gloo IGMC_LOCAL_DEVICE=0 timeout 600 python bench.py --gpus $n --steps 20 --warmup 5 --profile-steps 0 --rmse-links 0 \
    > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err
  echo "rc=$?"
  python - "$O/bench_${n}ranks.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['n_gpus'], 'ranks:', round(d['value']), 'sg/s', json.dumps(d['dp_check']))
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
done
unset IGMC_GRAPH_STEP
# refusal: two ranks asked for, one device, no dry-run request
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_refused.json 2> $O/bench_refused.err; echo "refusal rc=$? stdout bytes=$(wc -c < $O/bench_refused.json)"; tail -1 $O/bench_refused.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_driver.json 2> $O/bench_driver.err
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_d*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2), 'us/step frac', r.get('frac'), 'floor', r.get('floor_us'), 'eval', (d.get('rmse') or {}).get('eval_subgraphs_per_s'))
        print('   kernels', d.get('kernels_us'))
        for k,v in (d.get('secondary') or {}).items(): print('   ', k, v.get('value'), v.get('us_per_step'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-500:])
PY
