#!/bin/bash
# the reference's recipe on the three bundled Monti datasets (README of the reference: 40 epochs here, testing + ensemble),
# run from a scratch directory; the logs go to gpurun_out/<tag>/
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-recipes}; mkdir -p $O
export TMPDIR=/tmp
S=/tmp/igmc_recipes; rm -rf $S; mkdir -p $S; cd $S
for d in douban flixster yahoo_music; do
  t0=$(date +%s)
  { echo "# commit ${IGMC_COMMIT:-unknown}; cwd scratch; python $ROOT/Main.py --data-name $d --epochs 40 --testing --ensemble"; timeout 900 python $ROOT/Main.py --data-name $d --epochs 40 --testing --ensemble 2>&1; } > $O/recipe_$d.log
  echo "$d rc=$? $(( $(date +%s) - t0 )) s: $(tail -1 $O/recipe_$d.log)"
done
