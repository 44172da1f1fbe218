#!/bin/bash
# BASELINE config 5 end to end (reference README.md:51 + run_transfer_exps.sh:9-22): the ml_100k recipe -- max-nodes-per-hop 200,
# 80 epochs, --dynamic-train --testing --ensemble (synthetic ml_100k-shaped ratings: MovieLens is not available offline) -- then
# the TRANSFER ensemble evaluation of its checkpoints 10..40 on the three bundled Monti datasets (--num-relations 5; yahoo_music
# predictions x 20).  Run from a scratch directory; logs to gpurun_out/<tag>/ with wall times and evaluation rates.
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-config5}; mkdir -p $O
export TMPDIR=/tmp
S=/tmp/igmc_config5; rm -rf $S; mkdir -p $S; cd $S
ln -s $ROOT/raw_data raw_data 2>/dev/null
t0=$(date +%s.%N)
CMD="python $ROOT/Main.py --data-name ml_100k --save-appendix _mnph200 --data-appendix _mnph200 --epochs 80 --max-nodes-per-hop 200 --testing --ensemble --dynamic-train"
{ echo "# commit ${IGMC_COMMIT:-unknown}; cwd scratch; $CMD"; timeout 1500 $CMD 2>&1; } > $O/recipe_ml_100k_mnph200_80epochs.log
echo "ml_100k rc=$? $(python -c "import time;print('%.1f s' % (time.time()-$t0))"): $(tail -1 $O/recipe_ml_100k_mnph200_80epochs.log)"
for d in douban flixster yahoo_music; do
  mb=1; [ $d = yahoo_music ] && mb=20
  t0=$(date +%s.%N)
  CMD="python $ROOT/Main.py --data-name $d --epochs 40 --testing --no-train --ensemble --transfer results/ml_100k_mnph200_testmode/ --num-relations 5 --multiply-by $mb"
  { echo "# commit ${IGMC_COMMIT:-unknown}; cwd scratch; $CMD"; timeout 600 $CMD 2>&1; } > $O/transfer_$d.log
  echo "$d rc=$? $(python -c "import time;print('%.1f s' % (time.time()-$t0))"): $(grep 'Test Once' $O/transfer_$d.log | tail -1) | $(tail -1 $O/transfer_$d.log)"
done
