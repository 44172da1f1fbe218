"""Experiment (GPU, variant library with stamps in k_finalize_ts: profiles/r06_experiments): cycles between the stamps of a main
workgroup (layer 1, part 0), a lin workgroup and the loss / tick workgroup of the LAST k_finalize_ts launch of a short run."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'igmc_amd', 'lib', 'libigmc_hip_finclk.so')
import torch
from igmc_amd import preprocessing
from igmc_amd.models import IGMC
from igmc_amd.stepgraph import StepGraph
from igmc_amd.train_eval import FlatAdam
from igmc_amd.util_functions import MyDynamicDataset
split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
ds = MyDynamicDataset('data/x', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, cv, device=0, seed=1)
model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
model.reset_parameters()
opt = FlatAdam(model, lr=1e-3)
sg = StepGraph(model, opt, ds, 50, 0.001)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))
sg.begin_epoch(perm, 1)
sg.steps(1)
sg.prepare(steps_hint=64)
for rep in range(3):
    sg.steps(64)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 64)()
    lib = _lib.load()
    lib.cdll.igmc_debug_fin_clocks.argtypes = [C.c_void_p]
    assert lib.cdll.igmc_debug_fin_clocks(out) == 0
    v = list(out)
    t0 = v[0]
    def row(base, ks, names):
        return '  '.join('%s +%d' % (n, v[base + k] - v[base]) for k, n in zip(ks, names))
    print('main wg (layer 1, part 0): ' + row(0, [1, 2, 3, 4, 5, 6, 7], ['loads issued', 'stash in LDS', 'barrier', 'main pass stored', 'roles done', 'img barrier', 'images stored']))
    print('lin wg: start +%d  ' % (v[16] - t0) + row(16, [1, 7], ['begin', 'adam done']))
    print('tick wg: start +%d  ' % (v[32] - t0) + row(32, [1, 2, 7], ['begin', 'loss done', 'tick done']))
