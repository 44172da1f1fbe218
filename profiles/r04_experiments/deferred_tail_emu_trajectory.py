"""Deferred tail on the emulator: K train steps with the tail of step t inside the subgraph launch of step t + 1 must leave
the same bits as the plain sequence (same kernels, same order of operations; only the launch that carries the tail moves)."""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ctypes as C
import numpy as np
import parity_checks as PC
from helpers import load_extract_golden
from igmc_amd import engine
import test_gpu_headline as H

def run(be, case, R, steps, B, defer, use_dropout=False, ctrl=False):
    L = 2 * case['h'] + 2
    g = engine.Graph(case['A'], device=be.device, lib=be.lib)
    arenas = [engine.Batch(g, max_graphs=B, hop=case['h'], max_nodes_per_hop=case['mnph']) for _ in range(steps)]
    ys = case['class_values'][case['link_labels']].astype(np.float32)
    lu, lv, ly = be.dev(case['links'][:, 0].astype(np.int32)), be.dev(case['links'][:, 1].astype(np.int32)), be.dev(ys)
    b0 = arenas[0]
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, 0, b0.node_capacity, b0.edge_capacity, B)
    ref = PC.make_ref_model(L, R, seed=4)
    n_p = ws.n_params
    P = be.dev(PC.flatten_params(ws, ref))
    M1, M2, G = be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32))
    out, loss, total = be.dev(np.zeros(B, np.float32)), be.dev(np.zeros(2, np.float32)), be.dev(np.zeros(1, np.float64))
    rng = np.random.default_rng(4)
    cd = be.lib.cdll
    cd.igmc_model_defer_tail.argtypes = [C.c_void_p, C.c_int]
    cd.igmc_model_flush_tail.argtypes = [C.c_void_p, C.c_void_p]
    if defer:
        assert cd.igmc_model_defer_tail(ws.handle, 1) == 0
    for s in range(steps):
        b = arenas[s]
        b.extract(be.ptr(lu), be.ptr(lv), be.ptr(ly), None, s * B, B, sample_ratio=case['sample_ratio'], seed=2, epoch=1)
    losses = []
    for s in range(steps):
        b = arenas[s]
        lm = rng.random((B, 128)) < 0.5
        LM = be.dev(lm.astype(np.uint8).reshape(-1))
        if s > 0:
            be.lib.call('igmc_model_weights_unchanged', ws.handle, 1)
        be.lib.call('igmc_train_step', ws.handle, engine._p(be.ptr(P)), b.handle, int(use_dropout), engine._p(be.ptr(LM)),
                    0, 0, 1.0, 0.001, engine._p(be.ptr(out)), engine._p(be.ptr(G)), engine._p(be.ptr(M1)),
                    engine._p(be.ptr(M2)), engine._p(be.ptr(loss)), engine._p(be.ptr(total)), None, s + 1, 1e-3, 0.9, 0.999,
                    1e-8, 0.0, None)
        be.sync()
        losses.append(be.host(loss).copy())
    if defer:
        assert cd.igmc_model_flush_tail(ws.handle, None) == 0
    be.sync()
    be.lib.call('igmc_model_check', ws.handle, None)
    return be.host(P).copy(), be.host(M1).copy(), be.host(M2).copy(), be.host(loss).copy(), be.host(total).copy(), losses

if __name__ == '__main__':
    be = PC.EmuBackend()
    import os
    print('lib:', be.lib.cdll._name)
    os.environ['IGMC_GRAPH_STEP'] = '1'
    os.environ['IGMC_GS_TRACE'] = '1'
    CASES = load_extract_golden()
    for name, R, B, steps in (('douban', 5, 4, 4), ('synth_cap', 5, 4, 3)):
        case = CASES[name]
        a = run(be, case, R, steps, B, defer=False)
        d = run(be, case, R, steps, B, defer=True)
        same = all(np.array_equal(x, y) for x, y in zip(a[:5], d[:5]))
        print(name, 'params/moments/loss/total identical:', same, 'final loss', a[3], d[3])
        print('  per-step loss reads  plain:', [float(l[0]) for l in a[5]], ' deferred (one launch late):', [float(l[0]) for l in d[5]])
        assert same
