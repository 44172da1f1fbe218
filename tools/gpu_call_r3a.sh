#!/bin/bash
# round 3, GPU session A: suite, reproducibility comparisons, bench lines, kernel trace
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -25 $O/pytest.log
timeout 600 python tools/exp_repro.py 20 > $O/repro.txt 2>&1; echo "repro rc=$?" | tee -a $O/repro.txt; grep "==\|DIFF\|RAISED\|Error" $O/repro.txt | head -20
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err; done
timeout 400 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 > $O/bench_200b.json 2> $O/bench_200b.err
timeout 300 python bench.py --config douban --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_douban.json 2> $O/bench_douban.err
timeout 300 python bench.py --config ml_100k --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_ml100k.json 2> $O/bench_ml100k.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,1),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), d.get('timing_check'), d.get('dp_structure_us'), d.get('kernels_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-1500:])
PY
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o trace -- python $OLDPWD/bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof_bench.err )
python tools/rocprof_summary.py $O/prof > $O/kernel_stats.txt 2>&1; head -40 $O/kernel_stats.txt
