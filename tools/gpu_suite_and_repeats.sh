#!/bin/bash
# full GPU suite, then the cap-200 fused-step tests (oracle tracking + bit reproducibility) repeated in both finalize modes
set -u
ROOT=$(pwd); export PYTHONPATH=$ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; echo "full suite rc=$?"; tail -3 gpurun_out/gpu_tests.log
for mode in 1 0; do
  f=0
  for i in 1 2 3 4; do
    IGMC_FIN_MODE=$mode timeout 300 python -m pytest tests/test_gpu_headline.py -m gpu -q -x -k "cap200_fused" > /tmp/t.log 2>&1 || { f=$((f+1)); grep -E "AssertionError|assert " /tmp/t.log | head -3; }
  done
  echo "IGMC_FIN_MODE=$mode failures: $f / 4"
done
