"""Deferred tail under the device-side step control (groups of M steps, edge dropout, tick inside the fused launch): the
trajectory, the control block and the epoch total must equal the plain sequence bit for bit."""
import os, sys
os.environ['IGMC_GS_CLUSTER'] = '4'
os.environ['IGMC_GRAPH_STEP'] = '1'
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ctypes as C
import numpy as np
import parity_checks as PC
from helpers import load_extract_golden
from igmc_amd import engine, _lib
from test_emu_ctrl import group_ctrl_words

CASES = load_extract_golden()
be = PC.EmuBackend()
lib = be.lib
print('lib:', lib.cdll._name)
cd = lib.cdll
cd.igmc_model_defer_tail.argtypes = [C.c_void_p, C.c_int]
cd.igmc_model_flush_tail.argtypes = [C.c_void_p, C.c_void_p]
case = CASES['synth_cap']
g = engine.Graph(case['A'], lib=lib)
n = len(case['links'])
lu = case['links'][:, 0].astype(np.int32).copy()
lv = case['links'][:, 1].astype(np.int32).copy()
ly = case['class_values'][case['link_labels']].astype(np.float32)
B, M, T = 2, 3, 9
rng = np.random.default_rng(1)
perm = np.concatenate([rng.permutation(n) for _ in range(-(-B * (T + 2 * M) // n))]).astype(np.int32)
sets = [[engine.Batch(g, B, 1, case['mnph']) for _ in range(M)] for _ in range(2)]
ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, sets[0][0].node_capacity, sets[0][0].edge_capacity, B)
P0 = PC.flatten_params(ws, PC.make_ref_model(4, 5, seed=4))


def run(defer, drop):
    P, M1, M2 = P0.copy(), np.zeros_like(P0), np.zeros_like(P0)
    G, out, loss = np.zeros_like(P0), np.zeros(B, np.float32), np.zeros(2, np.float32)
    total = np.zeros(1, np.float64)
    ctrl = group_ctrl_words(step=11, epoch=3, adam_t=1, batch=B, group=M)
    cp = C.c_void_p(ctrl.ctypes.data)
    for s in sets:
        for a in s:
            lib.call('igmc_batch_set_ctrl', a.handle, cp)
            a.set_lean(True)
    lib.call('igmc_model_set_ctrl', ws.handle, cp)
    bsets = [engine.BatchSet(s) for s in sets]
    bsets[0].extract(M, lu, lv, ly, perm, 0, B, 1.0, 7, drop_p=0.2 if drop else 0.0, drop_seed=7)
    gq, t, recs = 0, 0, []
    while t < T:
        bsets[1 - gq].extract(M, lu, lv, ly, perm, 1 - gq, B, 1.0, 7, drop_p=0.2 if drop else 0.0, drop_seed=7)
        if defer:
            lib.call('igmc_model_defer_tail', ws.handle, 1)
        for i in range(M):
            if i > 0:
                lib.call('igmc_model_weights_unchanged', ws.handle, 1)
            lib.call('igmc_train_step', ws.handle, C.c_void_p(P.ctypes.data), sets[gq][i].handle, int(drop), None, 7, 0, 1.0, 0.001,
                     C.c_void_p(out.ctypes.data), C.c_void_p(G.ctypes.data), C.c_void_p(M1.ctypes.data),
                     C.c_void_p(M2.ctypes.data), C.c_void_p(loss.ctypes.data), C.c_void_p(total.ctypes.data), cp, 1,
                     1e-3, 0.9, 0.999, 1e-8, 0.0, None)
        if defer:
            lib.call('igmc_model_flush_tail', ws.handle, None)
            lib.call('igmc_model_defer_tail', ws.handle, 0)
        recs.append((P.copy(), loss.copy(), out.copy()))
        t += M
        gq ^= 1
    lib.call('igmc_model_check', ws.handle, None)
    for s in sets:
        for a in s:
            lib.call('igmc_batch_set_ctrl', a.handle, None)
            a.set_lean(False)
    lib.call('igmc_model_set_ctrl', ws.handle, None)
    return P, M1, M2, ctrl.copy(), total.copy(), recs


K = _lib.CTRL
for drop in (False, True):
    a = run(False, drop)
    d = run(True, drop)
    assert a[3][K['SYNC_ERR']] == 0 and d[3][K['SYNC_ERR']] == 0, (a[3][K['SYNC_ERR']], d[3][K['SYNC_ERR']])
    ok = all(np.array_equal(x, y) for x, y in zip(a[:5], d[:5]))
    okg = all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a[5], d[5]))
    print('dropout', drop, ': params / moments / control block / total identical:', ok, '; at every group end:', okg, 'step', d[3][K['STEP']], 'total', d[4])
    assert ok and okg
