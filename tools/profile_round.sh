#!/bin/bash
# Round-end measurement recipe (run on the GPU box through gpurun; writes under gpurun_out/prof/).
#   IGMC_COMMIT=<short hash> bash tools/profile_round.sh [config]
# Every summary starts with the exact command it came from.
set -u
CFG=${1:-ml_1m}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
BENCH="python $ROOT/bench.py --config $CFG --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary"
cd /tmp
# 1. kernel trace (graph replay + overlap)
timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/kt -- $BENCH > $ROOT/$OUT/kt.log 2>&1
{ echo "# commit ${IGMC_COMMIT:-unknown}; rocprofv3 --kernel-trace --stats -- $BENCH"; python $ROOT/tools/rocprof_summary.py $ROOT/$OUT/kt; } > $ROOT/$OUT/kernel_stats.txt 2>&1
# 2. PMC passes (own runs, kernel-trace only; rocprofv3 serialises the dispatches while it collects counters -- the grouped
#    step graph has no device-side waits between its two chains, so that changes timing only)
PB="python $ROOT/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT -d $ROOT/$OUT/pmc1 -- $PB > $ROOT/$OUT/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $ROOT/$OUT/pmc2 -- $PB > $ROOT/$OUT/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $ROOT/$OUT/pmc3 -- $PB > $ROOT/$OUT/pmc3.log 2>&1
for p in pmc1 pmc2 pmc3; do { echo "# commit ${IGMC_COMMIT:-unknown}; rocprofv3 --kernel-trace --pmc ... -- $PB (see tools/profile_round.sh)"; python $ROOT/tools/rocprof_summary.py $ROOT/$OUT/$p --pmc; } > $ROOT/$OUT/$p.txt 2>&1; done
rm -rf $ROOT/$OUT/kt $ROOT/$OUT/pmc1 $ROOT/$OUT/pmc2 $ROOT/$OUT/pmc3       # keep the summaries only (<= 64 MiB rule)
cd $ROOT
python $ROOT/tools/kernel_resources.py > $ROOT/$OUT/kernel_resources.txt 2>&1     # registers / LDS / waves per SIMD
python $ROOT/tools/pmc_traffic.py $ROOT/$OUT $ROOT/$OUT/pmc_traffic.json $CFG > /dev/null 2>&1
