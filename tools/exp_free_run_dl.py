"""Debug aid: free-running prefetch on the dense per-layer path (config 2 shape), bit-identity vs fork / join
(GPU run of round 2: identical parameters over two epochs of 20 steps; bench 262-265 k -> 272.6 k subgraphs/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from igmc_amd import preprocessing
from igmc_amd.models import IGMC
from igmc_amd.stepgraph import StepGraph
from igmc_amd.train_eval import FlatAdam
from igmc_amd.util_functions import MyDynamicDataset
split = preprocessing.create_trainvaltest_split('ml_100k', 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
pick = np.random.default_rng(3).permutation(len(tr_u))[:1000]
ds = MyDynamicDataset('data/t/frdl', A, (tr_u[pick], tr_v[pick]), np.asarray(tr_l)[pick], 1, 1.0, 200, None, None, cv, device=0, seed=1)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
out = {}
for mode in ('0', '1'):
    os.environ['IGMC_FREE_RUN'] = mode
    torch.manual_seed(3)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.2, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001)
    assert sg.arenas[0].dense_layers(sg.ws) and sg.free_run == (mode == '1'), (sg.free_run, mode)
    t1, _ = sg.run_epoch(perm, 1)
    t1 = float(t1.item())
    t2, _ = sg.run_epoch(perm, 2)
    torch.cuda.synchronize()
    out[mode] = (model.flat_parameters().detach().cpu().clone(), t1, float(t2.item()), sg.multi is not None)
print('identical params', torch.equal(out['0'][0], out['1'][0]), 'totals', out['0'][1:] , out['1'][1:])
