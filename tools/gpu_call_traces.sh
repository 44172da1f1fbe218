#!/bin/bash
# rocprofv3 kernel traces of the secondary configurations (no counters): per-kernel launch times under the bench's own command
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-traces}; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
run() {   # name, bench arguments
  local n=$1; shift
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/kt_$n -- python $R/bench.py "$@" --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary > $R/$O/kt_$n.log 2>&1 )
  { echo "# commit ${IGMC_COMMIT:-unknown}; rocprofv3 --kernel-trace --stats -- python bench.py $* --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 --rmse-links 0 --dp-steps 0 --no-secondary"; python tools/rocprof_summary.py $O/kt_$n | head -16; grep '^{"metric"' $O/kt_$n.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('# bench line of the traced run: %.0f subgraphs/s, %.1f us/step' % (d['value'], d['ms_per_step']*1e3))"; } > $O/kernel_stats_$n.txt 2>&1
  rm -rf $O/kt_$n
}
run flixster --config flixster
run ml_10m_lite --config ml_10m_lite
run yahoo_music --config yahoo_music
run dgcnn_douban --dgcnn-rs --config douban
head -8 $O/kernel_stats_*.txt
