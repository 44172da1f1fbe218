#!/bin/bash
# phase clocks of the dense-layer kernels + bench lines (flixster, ml_100k, ml_1m) + the gpu suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-wide2}; mkdir -p $O
export TMPDIR=/tmp
for w in 0 2; do timeout 200 python tools/dl_phase_clocks.py flixster $w 2>&1 | grep -v amdgpu.ids > $O/clk_flixster_wg$w.txt; done
timeout 200 python tools/dl_phase_clocks.py ml_100k 0 2>&1 | grep -v amdgpu.ids > $O/clk_ml100k_wg0.txt
cat $O/clk_flixster_wg0.txt $O/clk_ml100k_wg0.txt
for c in flixster ml_100k ml_1m; do
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/full_$c.json 2> $O/full_$c.err
  python - $O/full_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step']*1e3,1), d.get('kernels_us'))
PY
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
