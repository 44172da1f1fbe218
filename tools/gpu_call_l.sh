#!/bin/bash
# GPU call L2: where the per-step gap is -- fork/join of the extraction branch vs kernel chain.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/l
mkdir -p $O
export PYTHONPATH=$ROOT
run() {
  local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --dp-steps 0 --profile-steps 0 --rmse-links 0 $ARGS ) > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print('%-30s'%sys.argv[2], round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3))
except Exception as e:
    print('ERR', e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
ARGS=""
run base A=1
run noext_nooverlap_chain IGMC_EXP_NO_EXTRACT=1 IGMC_NO_OVERLAP=1
run noext_nooverlap_nochain IGMC_EXP_NO_EXTRACT=1 IGMC_NO_OVERLAP=1 IGMC_NO_CHAIN=1
run nooverlap_chain IGMC_NO_OVERLAP=1
run noext_chain_g16 IGMC_EXP_NO_EXTRACT=1 IGMC_NO_OVERLAP=1 IGMC_GRAPH_STEPS=16
run noext_nograph IGMC_EXP_NO_EXTRACT=1 IGMC_NO_OVERLAP=1 IGMC_NO_GRAPH=1
