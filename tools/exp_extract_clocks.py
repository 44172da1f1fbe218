"""Experiment (GPU, variant library with stamps in k_extract_nodes): cycles between the phases of the workgroups of link 7 (user
side | item side) of a two-batch extraction launch run ALONE (nothing beside it)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'igmc_amd', 'lib', 'libigmc_hip_exclk.so')
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
sg = StepGraph(model, opt, ds, 50, 0.001, group=8)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))
sg.begin_epoch(perm, 1)
sg.steps(1)
lib = _lib.load()
lib.cdll.igmc_debug_ex_clocks.argtypes = [C.c_void_p]
names = ['init + link', 'expand fringe', 'new & ~visited, counts', 'sample (radix select)', 'append fringe', 'prefix of selected', 'ranks -> slots', 'clear block']
for rep in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sg._extract_many(0, 2)              # one two-batch launch of k_extract_nodes_set + k_relm_set, alone on the chip
    e1.record()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 32)()
    assert lib.cdll.igmc_debug_ex_clocks(out) == 0
    v = list(out)
    for side, base in (('users', 0), ('items', 16)):
        print('%s: ' % side + ' | '.join('%s %d' % (n, v[base + k + 1] - v[base + k]) for k, n in enumerate(names)) + ' | total %d' % (v[base + 8] - v[base]))
    print('   launch of both kernels alone: %.1f us' % (e0.elapsed_time(e1) * 1e3))
