#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -X faulthandler bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 --rmse-links 0 > $O/bench_dp.json 2> $O/bench_dp.err; echo "dp leg rc=$?"; tail -25 $O/bench_dp.err
timeout 300 python -X faulthandler bench.py --steps 200 --warmup 20 --no-cpu-baseline --profile-steps 0 --dp-steps 0 > $O/bench_rmse.json 2> $O/bench_rmse.err; echo "rmse leg rc=$?"; tail -25 $O/bench_rmse.err
timeout 300 python -X faulthandler bench.py --steps 200 --warmup 20 --no-cpu-baseline --rmse-links 0 --dp-steps 0 > $O/bench_prof.json 2> $O/bench_prof.err; echo "profile leg rc=$?"; tail -25 $O/bench_prof.err
