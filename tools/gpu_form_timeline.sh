#!/bin/bash
# dispatch timeline of the driver's 20-step form (one graph launch after a synchronize): where its microseconds over the
# 200-step form go -- in front of the first kernel, or inside the launch
#   gpurun -- 'bash tools/gpu_form_timeline.sh <tag>'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-form_timeline}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$O/kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary --no-floor > $R/$O/kt.log 2>&1 )
python tools/rocprof_summary.py $O/kt --timeline > $O/timeline_all.txt 2>&1
tail -700 $O/timeline_all.txt > $O/timeline_tail.txt
grep '^{"metric"' $O/kt.log | tail -1 > $O/bench_traced.json
rm -rf $O/kt
wc -l $O/timeline_all.txt; tail -5 $O/timeline_tail.txt
