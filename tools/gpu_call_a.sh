#!/bin/bash
# GPU call A (round 2): new parity tests, bench lines, full-recipe training logs.  Writes under gpurun_out/a/.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/a
mkdir -p $O
export PYTHONPATH=$ROOT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > $O/gpu_tests.log
( timeout 300 python bench.py --steps 20 --warmup 5 ) > $O/bench_ml1m_driver.json 2> $O/bench_ml1m_driver.err
( timeout 300 python bench.py --no-cpu-baseline ) > $O/bench_ml1m_200.json 2> $O/bench_ml1m_200.err
( timeout 300 python bench.py --config ml_100k ) > $O/bench_ml100k.json 2> $O/bench_ml100k.err
( timeout 300 python bench.py --config douban ) > $O/bench_douban.json 2> $O/bench_douban.err
# full recipe of the reference README (Flixster / Douban / YahooMusic): 40 epochs, LR decay 50, ensemble of 10/20/30/40
for d in flixster douban yahoo_music; do
  W=/tmp/recipe_$d; rm -rf $W; mkdir -p $W; cd $W
  CMD="python $ROOT/Main.py --data-name $d --epochs 40 --testing --ensemble"
  echo "# commit ${IGMC_COMMIT:-unknown}; cwd scratch; $CMD" > $O/recipe_$d.log
  ( timeout 600 $CMD 2>&1 | grep -v "^Saving" | tail -60 ) >> $O/recipe_$d.log
  cat $W/results/${d}_testmode/log.txt >> $O/recipe_$d.log 2>/dev/null
  cd $ROOT
done
tail -3 $O/gpu_tests.log
for f in $O/bench_*.json; do echo $f; cut -c1-400 $f; done
for d in flixster douban yahoo_music; do tail -2 $O/recipe_$d.log; done
