"""Experiment (GPU): what does the FIRST replay of a freshly captured hipGraph cost, and does hipGraphUpload remove it?
    python tools/exp_graph_launch.py
"""
import ctypes
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

split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
torch.cuda.set_device(0)
ds = MyDynamicDataset('data/x', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, cv, device=0, seed=1)
model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
model.reset_parameters()
opt = FlatAdam(model, lr=1e-3)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))


def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e6


for upload in (False, True):
    sg = StepGraph(model, opt, ds, 50, 0.001)
    sg.begin_epoch(perm, 1)
    for _ in range(2):
        sg.step()                      # eager (steps_done < 4)
    t_cap = timed(sg.prepare)
    if upload:
        hip = ctypes.CDLL('libamdhip64.so')
        st = torch.cuda.current_stream().cuda_stream
        for g in sg.graphs + [sg.multi]:
            ex = g.raw_cuda_graph_exec()
            rc = hip.hipGraphUpload(ctypes.c_void_p(ex), ctypes.c_void_p(st))
            print('hipGraphUpload rc', rc)
        torch.cuda.synchronize()
    out = []
    # k == 2 now: multi ok
    for i in range(4):
        out.append(('multi8', timed(lambda: sg.steps(8))))
    for i in range(6):
        out.append(('single', timed(lambda: sg.step())))
    print('upload=%s capture %.0f us' % (upload, t_cap), ' '.join('%s:%.0f' % x for x in out))
    sg.detach()
