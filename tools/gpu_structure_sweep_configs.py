"""GPU check, companion of tools/gpu_structure_sweep.py: the same comparison (replayed launches vs eager launches over three
epochs with ragged last batches) on the other step forms -- flixster (group-split dense-layer kernels, gated extraction, batch
by batch), the MovieLens-1M shape (subgraph kernel, sampled extraction), the ml_100k shape (dense-layer kernels, free chain).

    python tools/gpu_structure_sweep_configs.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from igmc_amd import preprocessing
from igmc_amd.util_functions import MyDynamicDataset
from igmc_amd.models import IGMC
from igmc_amd.stepgraph import StepGraph
from igmc_amd.train_eval import FlatAdam


def trajectory(ds, drop, perm, R, epochs, **sg_kw):
    torch.manual_seed(3)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=R, num_bases=4, regression=True, adj_dropout=drop, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, **sg_kw)
    totals = []
    for ep in range(1, epochs + 1):
        t, n = sg.run_epoch(perm, ep)
        totals.append(float(t.item()))
    torch.cuda.synchronize()
    return sg, (model.flat_parameters().detach().cpu().clone(), opt.exp_avg.detach().cpu().clone(), opt.exp_avg_sq.detach().cpu().clone(), totals, opt.t)


bad = 0
for name, mnph in (('flixster', 10000), ('ml_1m', 100), ('ml_100k', 200)):
    if name == 'flixster':
        (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti(name, testing=True)
    else:
        (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.create_trainvaltest_split(name, 1234, True, verbose=False)
    for n, M, drop in ((50 * 57 + 13, 20, 0.2), (50 * 101 + 49, 50, 0.0), (50 * 33 + 9, 8, 0.2)):
        ds = MyDynamicDataset('data/t/w_%s_%d' % (name, n), A, (tr_u[:n], tr_v[:n]), tr_l[:n], 1, 1.0, mnph, None, None, cv, device=0, seed=1)
        perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
        sg, ref = trajectory(ds, drop, perm, len(cv), 3, group=M)
        _, eager = trajectory(ds, drop, perm, len(cv), 3, use_graph=False, overlap=False, group=M)
        same = all((torch.equal(x, y) if torch.is_tensor(x) else x == y) for x, y in zip(ref, eager))
        bad += not same
        print(name, n, 'links', n // 50, 'steps +', n % 50, 'M', M, 'drop', drop, 'form', sg._step_form(), 'graphs', [g is not None for g in sg.graphs], 'SAME' if same else 'DIFFERENT', flush=True)
print('all structures agree' if not bad else '%d DIFFER' % bad)
