#!/bin/bash
# free-running prefetch (IGMC_FREE_RUN): GPU suite (incl. the bit-identity test), bench both ways, PMC passes 1+2 of the
# new sources (fork/join structure: rocprofv3 serialises dispatches while it collects counters)
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/fr; mkdir -p $O $ROOT/gpurun_out/prof; export PYTHONPATH=$ROOT
timeout 600 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -4 $O/gpu_tests.log
run() {
  local name=$1; shift
  ( env "$@" timeout 200 python bench.py --no-cpu-baseline --dp-steps 0 --rmse-links 0 ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d['roofline']
    print('%-10s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), 'g2 avg_us %.1f'%r['avg_us'], 'frac %.3f'%r['frac'], d.get('timing_check'))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
run free IGMC_FREE_RUN=1
run forkjoin IGMC_FREE_RUN=0

export TMPDIR=/tmp IGMC_FREE_RUN=0
PB="python $ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT -d $ROOT/gpurun_out/prof/pmc1 -- $PB > $ROOT/gpurun_out/prof/pmc1.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $ROOT/gpurun_out/prof/pmc2 -- $PB > $ROOT/gpurun_out/prof/pmc2.log 2>&1
for p in pmc1 pmc2; do { echo "# commit ${IGMC_COMMIT:-unknown}; IGMC_FREE_RUN=0 rocprofv3 --kernel-trace --pmc ... -- $PB (tools/gpu_free_run.sh)"; python $ROOT/tools/rocprof_summary.py $ROOT/gpurun_out/prof/$p --pmc; } > $ROOT/gpurun_out/prof/$p.txt 2>&1; done
rm -rf $ROOT/gpurun_out/prof/pmc1 $ROOT/gpurun_out/prof/pmc2
cd $ROOT
python tools/pmc_traffic.py gpurun_out/prof gpurun_out/prof/pmc_traffic.json ml_1m | tail -1
