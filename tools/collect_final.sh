#!/bin/bash
# copies the outputs of a final GPU session (tools/gpu_call_final.sh <tag>) from gpurun_out/<tag>/ into profiles/ under the
# round's names:   bash tools/collect_final.sh r06_final3 r06
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/$1; R=${2:-r06}
for f in $S/bench_*.json; do n=$(basename $f .json); cp $f profiles/${R}_final_${n}.json; done
cp $S/kernel_stats.txt profiles/${R}_rocprofv3_kernel_stats.txt
cp $S/kernel_stats_ml100k.txt profiles/${R}_rocprofv3_kernel_stats_ml100k.txt
for p in pmc1 pmc2 pmc3; do cp $S/$p.txt profiles/${R}_rocprofv3_$p.txt; cp $S/${p}_ml100k.txt profiles/${R}_rocprofv3_${p}_ml100k.txt; done
cp $S/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $S/pmc_traffic_ml_100k.json profiles/${R}_pmc_traffic_ml_100k.json
cp $S/kernel_resources.txt profiles/${R}_kernel_resources.txt
cp $S/phase_clocks.txt profiles/${R}_g2_phase_clocks.txt
cp $S/phase_clocks_overlap.txt profiles/${R}_g2_phase_clocks_overlap.txt
cp $S/pytest.log profiles/${R}_final_gpu_tests.log
cp $S/smoke.log profiles/${R}_smoke.log
cp $S/dl_phase_clocks_flixster.txt profiles/${R}_dl_phase_clocks_flixster.txt
cp $S/dl_phase_clocks_ml100k.txt profiles/${R}_dl_phase_clocks_ml100k.txt
cp $S/sp_bwd_phase_clocks.txt profiles/${R}_sp_bwd_phase_clocks.txt
cp $S/sampler_stats.txt profiles/${R}_sampler_stats.txt
cp $S/eval_bench.txt profiles/${R}_eval_bench.txt
for f in $S/recipe_*.log $S/transfer_*.log; do cp $f profiles/${R}_$(basename $f); done
cp $S/recipes_summary.txt profiles/${R}_recipes_summary.txt
cp $S/dp_dry_runs.txt profiles/${R}_dp_dry_runs.txt
cp $S/parity_observed.txt profiles/${R}_parity_observed.txt
