#!/bin/bash
# start / end of every kernel dispatch of the headline bench under graph replay (gap analysis): gpurun_out/tl/timeline.txt
set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p $ROOT/gpurun_out/tl
BENCH="python $ROOT/bench.py --steps 64 --warmup 16 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 $*"
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/tl/kt -- $BENCH > $ROOT/gpurun_out/tl/kt.log 2>&1
python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/tl/kt --timeline > $ROOT/gpurun_out/tl/timeline.txt 2>&1
rm -rf $ROOT/gpurun_out/tl/kt
wc -l $ROOT/gpurun_out/tl/timeline.txt; tail -2 $ROOT/gpurun_out/tl/kt.log
