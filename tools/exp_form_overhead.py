"""Experiment (GPU): where do the microseconds of the driver's 20-step form (ONE graph launch of 20 steps between two
synchronizes: ~76 us/step) over the steady state (~71.7 us/step) go?

    python tools/exp_form_overhead.py [--group 10]

Timed by HIP events on the step stream, 12 regions each:
  idle      synchronize, then one graph launch (the driver's form)
  busy      synchronize, a ~500 us spin kernel, then the graph launch: its enqueue + launch latency hide behind the spin
  double    synchronize, then two graph launches back to back (the second one queued behind the first)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd import preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--group', type=int, default=10)
args = ap.parse_args()
M = args.group
split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
torch.cuda.set_device(0)
ds = MyDynamicDataset('data/x', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, cv, device=0, seed=1)
model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
model.reset_parameters()
opt = FlatAdam(model, lr=1e-3)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))
sg = StepGraph(model, opt, ds, 50, 0.001)
sg.begin_epoch(perm, 1)
sg.step()
sg.prepare(group=M)
sg.steps(4 * M)
torch.cuda.synchronize()


def region(kind):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    if kind == 'busy':
        torch.cuda._sleep(1200000)
    ev[1].record()
    sg.steps(2 * M * (2 if kind == 'double' else 1))
    t_enq = time.perf_counter() - t0
    ev[2].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    n = 2 * M * (2 if kind == 'double' else 1)
    return ev[1].elapsed_time(ev[2]) * 1e3 / n, ev[0].elapsed_time(ev[1]) * 1e3, wall * 1e6 / n, t_enq * 1e6


print('# M = %d (a graph launch = %d steps)' % (M, 2 * M))
for rep in range(3):
    for kind in ('idle', 'busy', 'double'):
        rs = [region(kind) for _ in range(12)]
        g = sorted(r[0] for r in rs)
        print('%-7s gpu us/step: median %.2f  min %.2f  max %.2f | in front %.0f us | wall us/step median %.2f | host enqueue %.0f us'
              % (kind, g[len(g) // 2], g[0], g[-1], rs[0][1], sorted(r[2] for r in rs)[6], sorted(r[3] for r in rs)[6]))
sg.check()
