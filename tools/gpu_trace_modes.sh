#!/bin/bash
# rocprofv3 kernel-trace average of the roofline kernel under the pacing modes of the extraction chain (same box):
#   gpurun -- 'bash tools/gpu_trace_modes.sh <tag> [reps] ["mode:delay ..."]'
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-trace_modes}; mkdir -p $O
REPS=${2:-2}; MODES=${3:-2:10 1:0 0:0}
export TMPDIR=/tmp
R=$PWD
for rep in $(seq 1 $REPS); do
  for mode in $MODES; do
    n=m${mode%%:*}_d${mode##*:}_$rep
    ( cd /tmp && env IGMC_EXTRACT_PACED=${mode%%:*} IGMC_GATE_DELAY_US=${mode##*:} timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$n -- python $R/bench.py --config ml_1m --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary > $R/$O/kt_$n.log 2>&1 )
    echo "$n: $(python tools/rocprof_summary.py $O/kt_$n | grep k_graph_step2 | head -1)  | bench $(grep '^{"metric"' $O/kt_$n.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.2f us/step" % (d["ms_per_step"]*1e3))')"
    rm -rf $O/kt_$n
  done
done
