#!/bin/bash
set -u
ROOT=$(pwd); O=$ROOT/gpurun_out/y; mkdir -p $O; export PYTHONPATH=$ROOT
for i in 1 2 3; do ( timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rmse-links 0 --dp-steps 0 ) 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('driver-form', round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), d.get('timing_check'))"; done
for i in 1 2; do ( timeout 100 python bench.py --no-cpu-baseline --rmse-links 0 --dp-steps 0 ) 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('200 steps  ', round(d['value']), 'us/step %.1f'%(d['ms_per_step']*1e3), d.get('timing_check'))"; done
