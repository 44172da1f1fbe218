#!/bin/bash
# same-box A/B of two builds of the library on other configurations: tools/gpu_ab_cfg.sh <tag> <config> [<config> ...]
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-abcfg}; mkdir -p $O; shift
export TMPDIR=/tmp
BASE=$PWD/igmc_amd/lib/libigmc_hip_base.so
for c in "$@"; do
  for i in 1 2; do
    for which in base new; do
      L=""; [ $which = base ] && L=$BASE
      IGMC_LIB_PATH=$L timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 --no-secondary --no-floor 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $which $i', round(d['value']), round(d['ms_per_step']*1e3,1))" | tee -a $O/ab.txt
    done
  done
done
