"""Debug aid: phase clocks of workgroup 0 of k_graph_step (IGMC_GS_TIMING=1) on the bench workload."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['IGMC_GS_TIMING'] = '1'
os.environ['IGMC_GRAPH_STEP'] = '1'
from igmc_amd import _lib, preprocessing  # noqa: E402
if os.environ.get('IGMC_LIB_PATH'):
    _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

NAMES = ['start', 'setup+order', 'L0 fwd', 'L1 fwd', 'L2 fwd', 'L3 fwd', 'head fwd', 'head bwd', 'L3 bwd', 'L2 bwd',
         'L1 bwd', 'L0 bwd', 'end']


def main():
    lib = _lib.load()
    split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, class_values) = split
    ds = MyDynamicDataset('data/bench', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, class_values, device=0, seed=1)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(class_values), num_bases=4, regression=True,
                 adj_dropout=0.0, multiply_by=1, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, use_graph=False, overlap=False)
    perm = torch.randperm(len(ds))[:5000]
    sg.begin_epoch(perm, 1)
    for _ in range(20):
        sg.step()
    torch.cuda.synchronize()
    buf = np.zeros(64, np.uint64)
    lib.cdll.igmc_debug_gs_clocks(C.c_void_p(buf.ctypes.data), 64)
    c = buf.astype(np.int64)
    info = sg.arena.info(torch.cuda.current_stream().cuda_stream)
    print('batch nodes %d edges %d' % (info.num_nodes, info.num_edges))
    for k in range(1, 13):
        print('%-12s %8d cycles' % (NAMES[k], c[k] - c[k - 1]))
    print('total        %8d cycles' % (c[12] - c[0]))
    print('L1 fwd, wave 0 first bundle: gather %d  mfma %d  epilogue %d | wave0 all bundles %d (layer start %d)' % (
        c[17] - c[16], c[18] - c[17], c[19] - c[18], c[20] - c[16], c[16] - c[2]))
    print('L3 bwd, wave 0 first bundle: hs+zero %d  gather %d  dX %d  wgrad %d  epilogue %d | wave0 all bundles %d '
          '(layer start %d)  reduce %d' % (c[25] - c[24], c[26] - c[25], c[27] - c[26], c[28] - c[27], c[29] - c[28],
                                          c[30] - c[24], c[24] - c[7], c[31] - c[30]))
    print('L1 fwd after the bundles: weights of the next layer requested %d | exchange + barrier %d' % (c[22] - c[20], c[3] - c[22]))
    print('L3 bwd after the bundles: barrier %d | table product %d | partial-table stores + barrier %d | next weights '
          'requested %d | exchange + barrier %d' % (c[13] - c[30], c[14] - c[13], c[15] - c[14], c[21] - c[15], c[31] - c[21]))
    _extra(c)
    print('setup: kernel start -> T0 staged %d | row starts/labels loaded %d | histogram %d | ranking %d | schedule %d | unit lists %d' % (
        c[56] - c[0], c[57] - c[56], c[58] - c[57], c[59] - c[58], c[60] - c[59], c[1] - c[60]))


def _extra(c):
    print('L1 fwd: waves finish their bundles at', [int(c[32 + w] - c[16]) for w in range(4)], '(from first bundle start of wave 0)')
    print('L3 bwd: waves finish their bundles at', [int(c[36 + w] - c[24]) for w in range(4)])
    t0 = min(int(c[40 + 4 * m]) for m in range(4) if c[40 + 4 * m] > 0) if any(c[40 + 4 * m] > 0 for m in range(4)) else 0
    for m in range(4):
        if c[40 + 4 * m] > 0:
            print('cluster member %d of subgraph 0: start %d  setup done %d  first exchange: rows published %d  all rows in LDS %d' % (
                m, c[40 + 4 * m] - t0, c[41 + 4 * m] - t0, c[42 + 4 * m] - t0, c[43 + 4 * m] - t0))


if __name__ == '__main__':
    main()

