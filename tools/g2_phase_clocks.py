"""Debug aid: phase clocks of k_graph_step2 (IGMC_GS_TIMING=1) on the bench workload: workgroup 0 (user side, member 0) and
member 2 (item side) of subgraph 0.   python tools/g2_phase_clocks.py [--overlap] [--group]
(--group: the clocks of the LAST step of a pair of groups launched eagerly -- what an experimental library with a deferred tail,
IGMC_LIB_PATH + IGMC_DEFER_TAIL=1, runs as a fused launch; its wait shows as stamps 55..57, its tail roles as slots 900..)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['IGMC_GS_TIMING'] = '1'
from igmc_amd import _lib, preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

NAMES = {1: 'T0 table staged', 2: 'labels + zero fills', 3: 'relm staged + one-hot planes', 4: 'A fragments built', 5: 'layer 0',
         6: 'L1 stage W', 7: 'L1 reload h0', 8: 'L1 compute + sync', 9: 'L2 stage W', 10: 'L2 reload h1', 11: 'L2 compute + sync',
         12: 'L3 stage W', 13: 'L3 reload h2', 14: 'L3 compute + sync', 15: 'readout polled', 16: 'head forward', 17: 'd feat',
         18: 'dPre3 set-up', 19: 'B3 stage W^T + bias', 20: 'B3 wave compute', 21: 'B3 sync', 22: 'B3 table product', 23: 'B3 reload dPre2',
         24: 'B2 stage W^T + bias', 25: 'B2 wave compute', 26: 'B2 sync', 27: 'B2 table product', 28: 'B2 reload dPre1',
         29: 'B1 stage W^T + bias', 30: 'B1 wave compute', 31: 'B1 sync', 32: 'B1 table product', 33: 'B1 sync', 34: 'layer-0 table',
         35: 'kernel end'}
ORDER = list(range(1, 36))


def main():
    if os.environ.get('IGMC_LIB_PATH'):
        _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
    lib = _lib.load()
    split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, class_values) = split
    ds = MyDynamicDataset('data/bench', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, class_values, device=0, seed=1)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(class_values), num_bases=4, regression=True,
                 adj_dropout=0.0, multiply_by=1, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    ov = '--overlap' in sys.argv
    gr = '--graph' in sys.argv          # the clocks of the LAST step of a hipGraph replay (what bench.py times)
    sg = StepGraph(model, opt, ds, 50, 0.001, use_graph=gr, overlap=ov or gr)
    perm = torch.randperm(len(ds))[:20000 if gr else 5000]
    sg.begin_epoch(perm, 1)
    for _ in range(20):
        sg.step()
    if gr:
        sg.prepare(steps_hint=200)
        sg.steps(200)
    if '--group' in sys.argv:
        sg.steps(2 * sg.M)
    torch.cuda.synchronize()
    buf = np.zeros(128, np.uint64)
    lib.cdll.igmc_debug_g2_clocks(C.c_void_p(buf.ctypes.data), 128)
    c = buf.astype(np.int64)
    print('overlap=%s; cycles per phase: member 0 (users) | member 2 (items); stamp 36..38 = wave 0 done with its L1..L3 bundle' % ov)
    for base, nm in ((0, 'member 0'), (64, 'member 2')):
        print(nm, 'start offset vs member 0: %d' % (c[base] - c[0]))
    for k in ORDER:
        a0 = c[k] - c[k - 1]
        a2 = c[64 + k] - c[64 + k - 1]
        print('%-28s %8d | %8d' % (NAMES[k], a0, a2))
    print('total                        %8d | %8d' % (c[35] - c[0], c[99] - c[64]))
    for l in (1, 2, 3):
        print('L%d fwd: wave 0 compute alone %d | %d' % (l, c[36 + l - 1] - c[7 + 3 * (l - 1)], c[100 + l - 1] - c[71 + 3 * (l - 1)]))
    fine(c)
    if c[55] and c[56] >= c[55]:          # (a library with the deferred tail: the wait for the previous step's tail)
        print('deferred tail: set-up done -> flag seen %d | fence + layer-0 table + step word %d  (cycles, member 0)' % (
            c[56] - c[55], c[57] - c[56]))
        big = np.zeros(1024 * 3, np.uint64)
        lib.cdll.igmc_debug_g2_wg_clocks(C.c_void_p(big.ctypes.data), 1024)
        bw = big.reshape(1024, 3).astype(np.int64)
        t0 = bw[:200, 0].min()
        roles = [i for i in range(900, 1024) if bw[i, 0] > 0]
        if roles:
            st, bd, ar = [(bw[roles, k] - t0) / 100.0 for k in (0, 2, 1)]
            print('tail roles (%d): start min/max %.1f / %.1f us | body done min/median/max %.1f / %.1f / %.1f | arrived max %.1f '
                  '(vs the first subgraph workgroup)' % (len(roles), st.min(), st.max(), bd.min(), np.median(bd), bd.max(), ar.max()))
            print('  per role body time (us): ' + ' '.join('%.1f' % x for x in (bd - st)))
    NWG = 224                     # 50 subgraphs in XCD-aligned blocks of 32 workgroups: wg = 32 j + 8 member + x, subgraph 8 j + x
    wg = np.zeros(NWG * 3, np.uint64)
    if hasattr(lib.cdll, 'igmc_debug_g2_wg_clocks'):
        lib.cdll.igmc_debug_g2_wg_clocks(C.c_void_p(wg.ctypes.data), NWG)
        w = wg.reshape(NWG, 3).astype(np.int64)
        sub = np.array([(i // 32) * 8 + (i % 8) for i in range(NWG)])
        mem = np.array([(i // 8) % 4 for i in range(NWG)])
        keep = sub < 50
        ids = np.arange(NWG)[keep]
        w, sub, mem = w[keep], sub[keep], mem[keep]
        t0 = w[:, 0].min()
        st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0          # us (100 MHz wall clock)
        print('per-workgroup (200 of 224): start min/median/max %.1f / %.1f / %.1f us; end min/median/max %.1f / %.1f / %.1f us; '
              'duration min/median/max %.1f / %.1f / %.1f us' % (st.min(), np.median(st), st.max(), en.min(), np.median(en),
                                                                 en.max(), (en - st).min(), np.median(en - st), (en - st).max()))
        late = np.argsort(-en)[:8]
        print('last to end: ' + ', '.join('wg %d (subgraph %d, member %d, xcc %d): start %.1f end %.1f' % (
            ids[i], sub[i], mem[i], w[i, 2] & 15, st[i], en[i]) for i in late))
        dmax = np.array([(en - st)[sub == g].max() for g in range(50)])
        print('per-subgraph duration of the slowest member: min %.1f median %.1f max %.1f us' % (dmax.min(), np.median(dmax), dmax.max()))
        xc = [len(set((w[sub == g, 2] & 15).tolist())) for g in range(50)]
        print('XCDs per cluster: max %d (1 = every cluster on one XCD)' % max(xc))


def fine(c):
    print('set-up (member 0): loop start -> loads issued %d | zero fills %d | label store %d | barrier %d' % (
        c[48] - c[1], c[49] - c[48], c[50] - c[49], c[2] - c[50]))
    print('layer 0: hist MFMAs %d | tile + transform %d | epilogue %d' % (c[51] - c[4], c[52] - c[51], c[5] - c[52]))
    print('L2 fwd wave 0: gather %d | transform %d | pair barrier %d | epilogue %d' % (c[40] - c[10], c[41] - c[40], c[42] - c[41], c[37] - c[42]))
    print('B2 wave 0: h loads issued %d | gather %d | tile + HS writes %d | transform %d | pair barrier %d | epilogue %d' % (
        c[43] - c[24], c[44] - c[43], c[45] - c[44], c[46] - c[45], c[47] - c[46], c[25] - c[47]))
    print('head: readout -> lin1 dot done %d | rest of head forward %d' % (c[54] - c[15], c[16] - c[54]))


if __name__ == '__main__':
    main()

