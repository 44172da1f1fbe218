"""flixster trajectory on the relation-group kernels: run-to-run bits, and the deviation from the oracle beside the
row-walker path's (IGMC_DL=0) on the same links."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np
import parity_checks as PC
import test_gpu_headline as T

be = PC.GpuBackend()
case = T.monti_case('flixster', 250)
def run(tag):
    try:
        r = PC.run_fused_train_trajectory(be, case, R=10, steps=5, batch=50, use_dropout=False)
        print(tag, 'ok frac_off %.3g max_diff %.3g' % (r['frac_off'], r['max_diff']))
        return r
    except AssertionError as e:
        print(tag, 'ASSERT', str(e)[:200])
        return None
import torch
print('torch threads', torch.get_num_threads())
a = run('wide 1'); b = run('wide 2')
if a and b:
    print('bit-identical params', np.array_equal(a['params'], b['params']), 'm1', np.array_equal(a['m1'], b['m1']))
torch.set_num_threads(1)
c = run('wide, oracle on 1 thread')
os.environ['IGMC_DL'] = '0'
d = run('row walkers, oracle on 1 thread')
torch.set_num_threads(8)
e = run('row walkers, 8 threads')
if a and c: print('wide vs wide(1 thread oracle) identical engine bits', np.array_equal(a['params'], c['params']))
for l in open(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'parity_observed.jsonl')):
    dd = json.loads(l)
    if dd.get('kind') == 'fused_trajectory': print({k: dd[k] for k in ('exp_avg_rel', 'exp_avg_sq_rel', 'params_frac_off', 'params_max_diff')})
