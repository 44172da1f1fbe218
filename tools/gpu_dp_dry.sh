#!/bin/bash
# dry runs of bench.py's N > 1 path with every rank on ONE GPU over the peer-mapped exchange (its throughput means nothing:
# the ranks share the chip; what counts is dp_check -- transport, ranks_agree, replicas_identical, allreduce_us)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/${1:-dpdry}; mkdir -p $O
export TMPDIR=/tmp
for n in 2 4 8; do
  # (more than two processes on one chip: the subgraph kernel's clusters of four co-resident workgroups per subgraph no longer
  #  fit next to each other -- its bounded polls raise; those runs take the per-layer kernels, the exchange is the same)
  [ $n -gt 2 ] && export IGMC_GRAPH_STEP=0
  # (round 6: no launcher -- bench.py spawns its N ranks itself; IGMC_LOCAL_DEVICE=0 is the explicit request for a one-GPU dry run)
  IGMC_DIST_BACKEND=gloo IGMC_LOCAL_DEVICE=0 timeout 600 python bench.py --gpus $n --dp-transport p2p --steps 20 --warmup 5 --profile-steps 0 --rmse-links 0 \
    > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err
  python - "$O/bench_${n}ranks.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(d['n_gpus'], 'ranks:', round(d['value']), 'sg/s', json.dumps(d['dp_check']))
except Exception as e:
    print(sys.argv[1], 'ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
