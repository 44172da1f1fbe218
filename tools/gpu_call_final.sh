#!/bin/bash
# Round-end GPU session: the profiling recipe first (kernel trace + PMC passes: bench.py's roofline.traffic is read from the
# summary of THIS kernel source), then the suite, smoke and the bench lines DESIGN.md quotes
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-final}; mkdir -p $O
export TMPDIR=/tmp
IGMC_COMMIT=${IGMC_COMMIT:-unknown} bash tools/profile_round.sh ml_1m > $O/profile_round.log 2>&1
cp gpurun_out/prof/*.txt gpurun_out/prof/*.json $O/ 2>/dev/null
cp gpurun_out/prof/pmc_traffic.json profiles/r06_pmc_traffic.json 2>/dev/null        # (same file is committed afterwards)
# config 2 (ml_100k, cap 200): the same recipe into its own directory, traffic record of its roofline kernel
mkdir -p gpurun_out/prof_ml1m && cp gpurun_out/prof/* gpurun_out/prof_ml1m/ 2>/dev/null
IGMC_COMMIT=${IGMC_COMMIT:-unknown} bash tools/profile_round.sh ml_100k > $O/profile_round_ml100k.log 2>&1
for f in pmc1 pmc2 pmc3; do cp gpurun_out/prof/$f.txt $O/${f}_ml100k.txt 2>/dev/null; done
cp gpurun_out/prof/pmc_traffic.json $O/pmc_traffic_ml_100k.json 2>/dev/null
cp gpurun_out/prof/pmc_traffic.json profiles/r06_pmc_traffic_ml_100k.json 2>/dev/null
cp gpurun_out/prof_ml1m/* gpurun_out/prof/ 2>/dev/null
# kernel trace of config 2 (ml_100k, cap 200)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt100k -- python $OLDPWD/bench.py --config ml_100k --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary > $OLDPWD/$O/kt100k.log 2>&1 )
{ echo "# commit ${IGMC_COMMIT:-unknown}; rocprofv3 --kernel-trace --stats -- python bench.py --config ml_100k --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary"; python tools/rocprof_summary.py $O/kt100k; } > $O/kernel_stats_ml100k.txt 2>&1
rm -rf $O/kt100k
rm -f gpurun_out/parity_observed.jsonl
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; cp gpurun_out/parity_observed.jsonl $O/parity_observed.jsonl 2>/dev/null; python tools/parity_observed_summary.py $O/parity_observed.jsonl "the -m gpu suite of this session, commit ${IGMC_COMMIT:-unknown}" > $O/parity_observed.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; head -4 $O/pytest.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err                      # the driver's default form (200 / 20, all legs)
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
# (the 20-step form varies by a few us/step from run to run: four more short ones, timed region only)
for i in 2 3 4 5; do
  timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor --profile-steps 0 > $O/bench_driver_$i.json 2> $O/bench_driver_$i.err
done
for c in douban ml_100k flixster ml_10m_lite yahoo_music; do
  st=200; [ $c = yahoo_music ] && st=64
  timeout 400 python bench.py --config $c --steps $st --warmup 20 --no-cpu-baseline --dp-steps 0 > $O/bench_$c.json 2> $O/bench_$c.err
done
timeout 300 python bench.py --dgcnn-rs --config douban --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 > $O/bench_dgcnn_douban.json 2> $O/bench_dgcnn_douban.err
timeout 200 python tools/g2_phase_clocks.py --graph > $O/phase_clocks.txt 2>&1
timeout 200 python tools/dl_phase_clocks.py flixster 0 2>&1 | grep -v amdgpu.ids > $O/dl_phase_clocks_flixster.txt
timeout 200 python tools/dl_phase_clocks.py ml_100k 0 2>&1 | grep -v amdgpu.ids > $O/dl_phase_clocks_ml100k.txt
timeout 200 python tools/sp_phase_clocks.py 2>&1 | grep -v amdgpu.ids | tail -11 > $O/sp_bwd_phase_clocks.txt
timeout 200 python tools/g2_phase_clocks.py --overlap > $O/phase_clocks_overlap.txt 2>&1
# round 6: the sampler's distribution + free-running statistical parity, evaluation throughput, the recipes (three Monti sets,
# BASELINE config 5), the data-parallel dry runs with self-spawned ranks on this one GPU
timeout 600 python tools/sampler_report.py > $O/sampler_stats.txt 2> $O/sampler_stats.err
{ python tools/eval_bench.py --links 20000; python tools/eval_bench.py --links 20000 --dynamic; python tools/eval_bench.py --links 5000; } 2>&1 | grep -v amdgpu.ids > $O/eval_bench.txt
IGMC_COMMIT=${IGMC_COMMIT:-unknown} bash tools/gpu_recipes.sh $(basename $O) > $O/recipes_summary.txt 2>&1
IGMC_COMMIT=${IGMC_COMMIT:-unknown} bash tools/gpu_recipes_config5.sh $(basename $O) >> $O/recipes_summary.txt 2>&1
bash tools/gpu_dp_dry.sh $(basename $O)/dp > $O/dp_dry_runs.txt 2>&1
cat $O/recipes_summary.txt; tail -3 $O/eval_bench.txt; tail -4 $O/sampler_stats.txt; cat $O/dp_dry_runs.txt | cut -c1-400
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d.get('roofline') or {}
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step', 'frac', r.get('frac'), 'avg_us', r.get('avg_us'), 'traffic', r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'), (d.get('rmse') or {}).get('value'), d.get('dp_structure_us'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
head -12 $O/kernel_stats.txt; head -14 $O/kernel_stats_ml100k.txt
