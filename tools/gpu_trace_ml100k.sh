#!/bin/bash
# kernel trace of BASELINE config 2 (was tools/gpu_call_s.sh) (ml_100k, cap 200) under graph replay
set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/s
BENCH="python $ROOT/bench.py --config ml_100k --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s/kt -- $BENCH > $ROOT/gpurun_out/s/kt.log 2>&1
{ echo "# commit ${IGMC_COMMIT:-unknown}; rocprofv3 --kernel-trace --stats -- $BENCH"; python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/s/kt; } > $ROOT/gpurun_out/s/kernel_stats_ml100k.txt 2>&1
rm -rf $ROOT/gpurun_out/s/kt
head -20 $ROOT/gpurun_out/s/kernel_stats_ml100k.txt; tail -2 $ROOT/gpurun_out/s/kt.log
