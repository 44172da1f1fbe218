#!/bin/bash
# same-box A/B: headline parity tests on the new build, bench base/new interleaved, phase clocks side by side
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-ab}; mkdir -p $O
export TMPDIR=/tmp
BASE=$PWD/igmc_amd/lib/libigmc_hip_base.so
timeout 600 python -m pytest tests/test_gpu_headline.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do
  IGMC_LIB_PATH=$BASE timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/base_$i.json 2> $O/base_$i.err
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --dp-steps 0 --rmse-links 0 --profile-steps 0 > $O/new_$i.json 2> $O/new_$i.err
done
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']), 'sg/s', round(d['ms_per_step']*1e3,2),'us/step')
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
IGMC_LIB_PATH=$BASE timeout 200 python tools/g2_phase_clocks.py > $O/pc_base.txt 2>&1
timeout 200 python tools/g2_phase_clocks.py > $O/pc_new.txt 2>&1
paste <(cut -c1-48 $O/pc_base.txt) <(cut -c29-48 $O/pc_new.txt) | sed -n 4,39p
grep "per-workgroup" $O/pc_base.txt $O/pc_new.txt
