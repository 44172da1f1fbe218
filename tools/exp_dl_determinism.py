"""Debug aid: is the dense per-layer path bit-reproducible?  Runs the cap-200 fused trajectory several times in one
process and compares the GPU results with each other (not with the oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import parity_checks as PC
import test_gpu_headline as T
be = PC.GpuBackend()
case = T.ml_case('ml_100k', 200, 50, seed=7)
import os
for env in ({},):
  os.environ.pop('IGMC_DL_DBG', None)
  os.environ.update(env)
  print('env', env)
  ref = None
  for i in range(int(os.environ.get('IGMC_EXP_RUNS', '6'))):
    try:
        r = PC.run_fused_train_trajectory(be, case, R=5, steps=1, batch=10, use_dropout=True)
        ok = 'oracle-ok'
    except AssertionError as e:
        print('run', i, 'oracle check failed:', str(e)[:100]); continue
    if ref is None:
        ref = r
    d = np.abs(ref['m1'] - r['m1'])
    lay = r['ws'].layout()
    worst = max(((float(d[o:o + int(np.prod(sh))].max()), k) for k, o, sh in lay))
    diffs = [(k, float(d[o:o + int(np.prod(sh))].max())) for k, o, sh in lay if d[o:o + int(np.prod(sh))].max() > 0]
    print('run', i, ok, 'm1 identical:', np.array_equal(ref['m1'], r['m1']), 'tensors that differ:', diffs)
