#!/bin/bash
# evaluation throughput: static (cached node sets) and dynamic test sets, kernel trace of the static one
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-evalb}; mkdir -p $O
export TMPDIR=/tmp
python tools/eval_bench.py --links 20000 2>&1 | grep -v amdgpu.ids | tee $O/static.txt
python tools/eval_bench.py --links 20000 --dynamic 2>&1 | grep -v amdgpu.ids | tee $O/dynamic.txt
python tools/eval_bench.py --links 5000 2>&1 | grep -v amdgpu.ids | tee $O/static5000.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt -- python $OLDPWD/tools/eval_bench.py --links 20000 > $OLDPWD/$O/kt.log 2>&1 )
python tools/rocprof_summary.py $O/kt 2>&1 | head -14 | tee $O/kernel_stats_eval.txt
rm -rf $O/kt
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "eval" 2>&1 | tail -3
