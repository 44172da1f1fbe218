"""Debug aid: phase clocks of k_sp_bwd (IGMC_SP_TIMING=1) on the DGCNN_RS bench workload (douban).
   python tools/sp_phase_clocks.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['IGMC_SP_TIMING'] = '1'
import torch  # noqa: E402
import bench  # noqa: E402
from igmc_amd import _lib, preprocessing  # noqa: E402
from igmc_amd.models import DGCNN_RS  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

NAMES = ['loads -> LDS', 'rank + max-pool', 'conv2 weight gradient', 'd pooled sequence (dzp)', 'max-pool / ReLU backward',
         'pooled rows -> LDS', 'conv1 weight gradient', 'd node states']


def main():
    lib = _lib.load()
    split = preprocessing.load_data_monti('douban', testing=True)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, class_values) = split
    ds = MyDynamicDataset('data/bench', A, (tr_u, tr_v), tr_l, 1, 1.0, 10000, None, None, class_values, device=0, seed=1)
    model = DGCNN_RS(ds, latent_dim=[32, 32, 32, 1], k=0.6, num_relations=len(class_values), num_bases=4, regression=True,
                     adj_dropout=0.2, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, use_graph=False, overlap=False)
    sg.begin_epoch(torch.randperm(len(ds))[:5000], 1)
    for _ in range(8):
        sg.step()
    torch.cuda.synchronize()
    buf = np.zeros(16, np.uint64)
    lib.cdll.igmc_debug_sp_clocks(C.c_void_p(buf.ctypes.data), 16)
    c = buf.astype(np.int64)
    print('k_sp_bwd, workgroup 0, shader cycles (k = %s):' % getattr(model, 'k', '?'))
    for i, nm in enumerate(NAMES):
        print('  %-28s %7d' % (nm, c[i + 1] - c[i]))
    print('  total                        %7d' % (c[8] - c[0]))


main()
