"""GPU check: epochs of several lengths (with ragged last batches) in groups of several sizes, three epochs each -- the replayed
launch structures (pairs of groups, single-group launches, eager leftovers) against eager launches, bit for bit.

    python tools/gpu_structure_sweep.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from igmc_amd import preprocessing
from igmc_amd.util_functions import MyDynamicDataset
import test_gpu_headline as H
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('douban', testing=True)
bad = 0
for n, M, drop in ((1456, 8, 0.2), (50 * 57 + 13, 20, 0.0), (50 * 101 + 49, 50, 0.2), (50 * 75 + 1, 50, 0.0), (50 * 33, 32, 0.2), (50 * 64 + 7, 13, 0.2),
                   (50 * 150 + 3, 50, 0.2), (50 * 99 + 25, 33, 0.0), (50 * 21 + 5, 4, 0.2), (50 * 260 + 17, 50, 0.2)):
    ds = MyDynamicDataset('data/t/z_%d' % n, A, (tr_u[:n], tr_v[:n]), tr_l[:n], 1, 1.0, 10000, None, None, cv, device=0, seed=1)
    perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
    sg, ref = H._trajectory(ds, drop, perm, epochs=3, group=M)
    _, eager = H._trajectory(ds, drop, perm, epochs=3, use_graph=False, overlap=False, group=M)
    same = all((torch.equal(x, y) if torch.is_tensor(x) else x == y) for x, y in zip(ref, eager))
    bad += not same
    print(n, 'links', n // 50, 'steps +', n % 50, 'M', M, 'drop', drop, 'graphs', [g is not None for g in sg.graphs], 'SAME' if same else 'DIFFERENT', flush=True)
print('all structures agree' if not bad else '%d DIFFER' % bad)
