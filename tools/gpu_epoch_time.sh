#!/bin/bash
# wall time of whole epochs of the headline configuration through the reference's entry point (Main.py: ml_1m-shaped synthetic
# ratings, --dynamic-train, max-nodes-per-hop 100, batch 50; 900 k training links, 100 k test links evaluated every epoch)
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
export TMPDIR=/tmp
S=/tmp/igmc_epochs; rm -rf $S; mkdir -p $S; cd $S
CMD="python $ROOT/Main.py --data-name ml_1m --epochs ${1:-3} --testing --dynamic-train --max-nodes-per-hop 100 --save-interval 10"
echo "# $CMD"
t0=$(date +%s.%N); $CMD 2>&1 | grep -v amdgpu.ids | grep "Epoch\|Final\|Duration\|rror\|Traceback" | tail -12; python -c "import time; print(\"wall of the whole command: %.1f s\" % (time.time() - $t0))"
