#!/bin/bash
# GPU call J: DGCNN_RS tests first, then the whole GPU suite; a short DGCNN_RS training run on douban.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/j
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests/test_gpu_dgcnn.py -m gpu -q -x 2>&1 | tail -30 ) > $O/gpu_dgcnn.log
tail -30 $O/gpu_dgcnn.log
( timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.log
tail -4 $O/gpu_tests.log
( timeout 900 python Main.py --data-name douban --epochs 6 --testing --dgcnn-rs --save-appendix _dgcnn 2>&1 | tail -12 ) > $O/main_dgcnn_douban.log
cat $O/main_dgcnn_douban.log
