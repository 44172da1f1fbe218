#!/bin/bash
# suite + the sort-pool family bench (douban, flixster) with the kernel split
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-dg}; mkdir -p $O
bash tools/gpu_suite.sh ${1:-dg}
for c in douban flixster; do
  timeout 300 python bench.py --dgcnn-rs --config $c --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --no-secondary --no-floor > $O/dgcnn_$c.json 2> $O/dgcnn_$c.err
  python - $O/dgcnn_$c.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d['value']), round(d['ms_per_step']*1e3,1), d.get('kernels_us') or d.get('roofline'))
PY
done
