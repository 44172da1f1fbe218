"""Debug aid: phase clocks of k_dl_fwd / k_dl_bwd (IGMC_DL_TIMING=<workgroup + 1>) on a bench workload.
   python tools/dl_phase_clocks.py [config] [workgroup]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cfgname = sys.argv[1] if len(sys.argv) > 1 else 'flixster'
wg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ['IGMC_DL_TIMING'] = str(wg + 1)
import bench  # noqa: E402
from igmc_amd import _lib, preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402


def main():
    lib = _lib.load()
    cfg = bench.CONFIGS[cfgname]
    if cfg['dataset'] in ('douban', 'flixster', 'yahoo_music'):
        split = preprocessing.load_data_monti(cfg['dataset'], testing=True)
    else:
        rmap = {float(i): i / 2.0 for i in range(1, 11)} if cfg['dataset'] == 'ml_10m_lite' else None
        split = preprocessing.create_trainvaltest_split(cfg['dataset'], 1234, True, rating_map=rmap, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, class_values) = split
    ds = MyDynamicDataset('data/bench', A, (tr_u, tr_v), tr_l, 1, 1.0, cfg['mnph'], None, None, class_values, device=0, seed=1)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(class_values), num_bases=4, regression=True,
                 adj_dropout=cfg['adj_dropout'], multiply_by=1, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, use_graph=False, overlap=False)
    perm = torch.randperm(len(ds))[:5000]
    sg.begin_epoch(perm, 1)
    ng = (len(class_values) + 4) // 5
    for _ in range(12):
        sg.step()
    torch.cuda.synchronize()
    buf = np.zeros(128, np.uint64)
    lib.cdll.igmc_debug_g2_clocks(C.c_void_p(buf.ctypes.data), 128)
    c = buf.astype(np.int64)
    print('%s, workgroup %d (subgraph %d), relation groups %d; shader cycles' % (cfgname, wg, wg // 4, ng))
    f = c[:40]
    print('k_dl_fwd: set-up %d | layer 0 %d' % (f[1] - f[0], f[2] - f[1]))
    if f[31]:
        print('  (set-up: -> zero fills done %d | block rows stored %d | barrier %d;  layer 0: histogram %d | pair barrier %d | table product + tanh %d | epilogue %d | end barrier %d)'
              % (f[31] - f[0], f[32] - f[31], f[1] - f[32], f[33] - f[1], f[34] - f[33], f[35] - f[34], f[36] - f[35], f[2] - f[36]))
    gs = ng == 2 and os.environ.get('IGMC_DL_GSPLIT', '1') != '0' and cfgname in ('flixster', 'ml_10m_lite')
    if gs:      # group split: both relation groups at once on the two halves of the workgroup (graphstep2.hip, GS)
        for l in (1, 2, 3):
            b = 3 + (l - 1) * 9
            print('L%d: reload + sync %d | gather %d | transform %d | partial hand-off + epilogue %d'
                  % (l, f[b + 1] - f[b], f[b + 3] - f[b + 1], f[b + 4] - f[b + 3], f[b + 8] - f[b + 4]))
        print('k_dl_fwd total %d' % (f[30] - f[0]))
        print('k_dl_bwd: set-up %d' % (c[42] - c[40]))
        for l in (3, 2, 1):
            k = 42 + (3 - l) * 14
            nxt = c[k + 14] if l > 1 else c[41]
            print('B%d: images / reload / sync %d | d bias %d | gather %d | transform %d | sync %d | epilogue + two table products %d | tail %d'
                  % (l, c[k + 1] - c[k], c[k + 2] - c[k + 1], c[k + 3] - c[k + 2], c[k + 4] - c[k + 3], c[k + 5] - c[k + 4],
                     c[k + 6] - c[k + 5], nxt - c[k + 6]))
        print('layer-0 table %d | k_dl_bwd total %d' % (c[126] - c[41], c[126] - c[40]))
        return
    for l in (1, 2, 3):
        b = 3 + (l - 1) * 9
        line = 'L%d: stage + reload + sync %d' % (l, f[b + 1] - f[b])
        prev = f[b + 1]
        for g in range(ng):
            s0, s1, s2 = f[b + 2 + 3 * g], f[b + 3 + 3 * g], f[b + 4 + 3 * g]
            line += ' | g%d: restage %d gather %d transform %d' % (g, s0 - prev, s1 - s0, s2 - s1)
            prev = s2
        line += ' | epilogue %d' % (f[b + 8] - prev)
        print(line)
    print('k_dl_fwd total %d' % (f[30] - f[0]))
    print('k_dl_bwd: set-up %d' % (c[42] - c[40]))
    for l in (3, 2, 1):
        for g in range(ng):
            k = 42 + ((3 - l) * ng + g) * 7
            d = [c[k + j + 1] - c[k + j] for j in range(6)]
            nxt = c[k + 7] if k + 7 < 42 + 3 * ng * 7 else c[41]
            print('B%d g%d: wpre/reload/stage/sync %d | d bias %d | gather %d | transform + epilogue %d | sync %d | tiles + table product %d | tail %d'
                  % (l, g, d[0], d[1], d[2], d[3], d[4], d[5], nxt - c[k + 6]))
    print('layer-0 table %d | k_dl_bwd total %d' % (c[126] - c[41], c[126] - c[40]))
    # every workgroup of the last k_dl_bwd launch on the 100 MHz wall clock
    # workgroups of a subgraph as graphstep2.hip's dl_split (DL_CAPS = the arena's slot capacities, users x items)
    cu, cv = [int(x) for x in os.environ.get('DL_CAPS', {'flixster': '50x155', 'ml_100k': '201x201', 'ml_10m_lite': '101x101'}.get(cfgname, '155x155')).split('x')]
    nqu, nqv = (cu + 127) // 128, (cv + 127) // 128
    nbu, nbv = (cu + 15) // 16, (cv + 15) // 16
    while nqu + nqv < 4:
        pu, pv = -(-nbu // nqu), -(-nbv // nqv)
        if max(pu, pv) <= 2:
            break
        if pu >= pv:
            nqu += 1
        else:
            nqv += 1
    per = nqu + nqv
    nwg = 50 * per
    wgb = np.zeros(3 * 1024, np.uint64)
    lib.cdll.igmc_debug_g2_wg_clocks(C.c_void_p(wgb.ctypes.data), 1024)
    w = wgb.reshape(1024, 3)[:nwg].astype(np.int64)
    t0 = w[:, 0].min()
    rows = []
    for i in range(nwg):
        if w[i, 1] == 0:
            continue
        rows.append(((w[i, 1] - t0) / 100.0, (w[i, 0] - t0) / 100.0, i, int(w[i, 2] >> 32), int(w[i, 2] & 0xFFFFFFFF)))
    rows.sort(reverse=True)
    print('k_dl_bwd workgroups with rows: %d of %d; end of the last one %.1f us after the first start' % (len(rows), nwg, rows[0][0]))
    for end, start, i, no, nop in rows[:8] + rows[len(rows) // 2:len(rows) // 2 + 3]:
        print('  wg %3d (subgraph %2d, member %d): start %.1f end %.1f us; own rows %d, opposite %d' % (i, i // per, i % per, start, end, no, nop))


main()
