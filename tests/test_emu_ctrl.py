"""Device-side step control (the hipGraph-replay mode): with the control block attached, the kernels must take
first / epoch / step / Adam scalars from HBM and produce exactly what the host-argument path produces
(kernel logic on the CPU emulation of the HIP sources)."""
import ctypes as C
import struct

import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden
from igmc_amd import _lib, engine

CASES = load_extract_golden()


def ctrl_words(step, first, epoch, adam_t, batch, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    w = np.zeros(_lib.CTRL['WORDS'], np.int64)
    w[0], w[1], w[2], w[3], w[4] = step, first, epoch, adam_t, batch
    w[6], w[7] = first + batch, 0          # odd slot, k
    for k, v in ((8, lr), (9, b1), (10, b2), (11, eps), (12, wd)):
        w[k] = struct.unpack('<q', struct.pack('<d', float(v)))[0]
    return w


@pytest.mark.parametrize('graph_step', ['0', '1', '1:4'])
def test_ctrl_path_equals_host_argument_path(monkeypatch, graph_step):
    graph_step, _, cluster = graph_step.partition(':')
    monkeypatch.setenv('IGMC_GRAPH_STEP', graph_step)     # per-layer kernels / one workgroup (or a cluster of 4) per subgraph
    if cluster:
        monkeypatch.setenv('IGMC_GS_CLUSTER', cluster)
    be = PC.EmuBackend()
    lib = be.lib
    case = CASES['synth_cap']
    A = case['A']
    g = engine.Graph(A, lib=lib)
    n = len(case['links'])
    lu = case['links'][:, 0].astype(np.int32).copy()
    lv = case['links'][:, 1].astype(np.int32).copy()
    ly = case['class_values'][case['link_labels']].astype(np.float32)
    perm = np.random.default_rng(0).permutation(n).astype(np.int32)
    B = 4
    b1 = engine.Batch(g, B, 1, case['mnph'])
    b2 = engine.Batch(g, B, 1, case['mnph'])
    ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, b1.node_capacity, b1.edge_capacity, B)
    ref = PC.make_ref_model(4, 5, seed=4)
    P0 = PC.flatten_params(ws, ref)
    outs = {}
    for mode in ('host', 'ctrl', 'finish', 'train_step'):
        P, M1, M2 = P0.copy(), np.zeros_like(P0), np.zeros_like(P0)
        G, out, loss = np.zeros_like(P0), np.zeros(B, np.float32), np.zeros(2, np.float32)
        batch = b1 if mode == 'host' else b2
        ctrl = ctrl_words(step=10, first=-2 * B, epoch=3, adam_t=0, batch=B)     # tick-before-use: slot[k&1] += 2B
        if mode in ('finish', 'train_step'):      # control block describes the NEXT step; the step's last kernel advances it
            ctrl = ctrl_words(step=11, first=0, epoch=3, adam_t=1, batch=B)
            for k, v in ((13, 1e-3 / (1 - 0.9)), (14, 1.0 / (1 - 0.999) ** 0.5)):
                ctrl[k] = struct.unpack('<q', struct.pack('<d', v))[0]
        total = np.zeros(1, np.float64)
        if mode != 'host':
            lib.call('igmc_batch_set_ctrl', batch.handle, C.c_void_p(ctrl.ctypes.data))
            lib.call('igmc_model_set_ctrl', ws.handle, C.c_void_p(ctrl.ctypes.data))
        else:
            lib.call('igmc_model_set_ctrl', ws.handle, None)
        rec = []
        for i in range(3):
            if mode == 'ctrl':
                lib.call('igmc_ctrl_tick', C.c_void_p(ctrl.ctypes.data), None)
                batch.extract(lu, lv, ly, perm, i & 1, B, 1.0, 7, 999)          # first = slot selector; epoch ignored
                batch.edge_dropout(0.2, False, 7, i & 1)                         # keyed by (epoch, batch index) of the slot
                ws.loss_grad(P.ctypes.data, batch, out.ctypes.data, G.ctypes.data, loss.ctypes.data,
                             use_edge_flags=True, seed=7, step=424242, ARR=0.001)
                lib.call('igmc_adam_step_ctrl', C.c_void_p(P.ctypes.data), C.c_void_p(G.ctypes.data),
                         C.c_void_p(M1.ctypes.data), C.c_void_p(M2.ctypes.data), len(P), C.c_void_p(ctrl.ctypes.data), None)
            elif mode == 'train_step':    # the whole step through igmc_train_step (multi-role launches + Adam tail)
                batch.extract(lu, lv, ly, perm, i & 1, B, 1.0, 7, 999)
                batch.edge_dropout(0.2, False, 7, i & 1)
                lib.call('igmc_train_step', ws.handle, C.c_void_p(P.ctypes.data), batch.handle, 1, None, 7, 0, 1.0, 0.001,
                         C.c_void_p(out.ctypes.data), C.c_void_p(G.ctypes.data), C.c_void_p(M1.ctypes.data),
                         C.c_void_p(M2.ctypes.data), C.c_void_p(loss.ctypes.data), C.c_void_p(total.ctypes.data),
                         C.c_void_p(ctrl.ctypes.data), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None)
            elif mode == 'finish':
                batch.extract(lu, lv, ly, perm, i & 1, B, 1.0, 7, 999)
                batch.edge_dropout(0.2, False, 7, i & 1)
                ws.loss_grad(P.ctypes.data, batch, out.ctypes.data, G.ctypes.data, None,
                             use_edge_flags=True, seed=7, step=424242, ARR=0.001)
                lib.call('igmc_step_finish', ws.handle, batch.handle, C.c_void_p(P.ctypes.data), C.c_void_p(G.ctypes.data),
                         C.c_void_p(M1.ctypes.data), C.c_void_p(M2.ctypes.data), 0.001, C.c_void_p(loss.ctypes.data),
                         C.c_void_p(total.ctypes.data), C.c_void_p(ctrl.ctypes.data), 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None)
            else:
                batch.extract(lu, lv, ly, perm, i * B, B, 1.0, 7, 3)
                batch.edge_dropout(0.2, False, 7, (3 << 32) ^ i)
                ws.loss_grad(P.ctypes.data, batch, out.ctypes.data, G.ctypes.data, loss.ctypes.data,
                             use_edge_flags=True, seed=7, step=11 + i, ARR=0.001)
                ws.adam_step(P.ctypes.data, G.ctypes.data, M1.ctypes.data, M2.ctypes.data, i + 1, 1e-3)
            d = batch.download()
            rec.append((d['node_gid'].copy(), d['eflag'].copy(), out.copy(), loss.copy(), P.copy()))
        outs[mode] = rec
        if mode == 'ctrl':
            assert ctrl[0] == 13 and ctrl[7] == 3 and ctrl[3] == 3
        if mode in ('finish', 'train_step'):
            assert ctrl[0] == 14 and ctrl[7] == 3 and ctrl[1] == 4 * B and ctrl[6] == 3 * B and ctrl[3] == 4 and ctrl[5] == 0
            assert total[0] == pytest.approx(sum(float(r[3][0]) * B for r in rec), rel=1e-6)
    for a, b in (list(zip(outs['host'], outs['ctrl'])) + list(zip(outs['host'], outs['finish'])) +
                 list(zip(outs['host'], outs['train_step']))):
        for x, y in zip(a[:2], b[:2]):
            assert np.array_equal(x, y)          # same links, same sampling, same dropout flags
        # lr is a float argument on the host path and a double in the control block: last-ulp differences only
        for x, y in zip(a[2:], b[2:]):
            np.testing.assert_allclose(x, y, rtol=2e-5, atol=1e-7)


def group_ctrl_words(step, epoch, adam_t, batch, group, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    from igmc_amd.stepgraph import _ctrl_words
    return _ctrl_words(step, epoch, adam_t, batch, group, lr, b1, b2, eps, wd)


def test_group_control_walks_the_same_batches():
    """Steps in GROUPS of M (igmc_hip.h, device-side step control; igmc_amd/stepgraph.py): the batches of a group sit in M
    arenas, the next group's are extracted through selectors q | (i << 1) into the other arena set -- before, between
    and after the steps of the running group, the cursor they resolve never moves meanwhile --, the tick of a group's
    last step moves that group's cursor on by two groups and flips the parity.  Same links, same sampling, same dropout
    draws, same parameters as the host-argument path, and no stamp mismatch; a batch extracted from the wrong cursor is
    caught by the consuming step's tick."""
    be = PC.EmuBackend()
    lib = be.lib
    case = CASES['synth_cap']
    g = engine.Graph(case['A'], lib=lib)
    n = len(case['links'])
    lu = case['links'][:, 0].astype(np.int32).copy()
    lv = case['links'][:, 1].astype(np.int32).copy()
    ly = case['class_values'][case['link_labels']].astype(np.float32)
    B, M, T = 2, 3, 8                                   # two full groups + two steps of the third
    rng = np.random.default_rng(1)      # (few links in the fixture: the epoch "permutation" visits them several times)
    perm = np.concatenate([rng.permutation(n) for _ in range(-(-B * (T + 2 * M) // n))]).astype(np.int32)
    sets = [[engine.Batch(g, B, 1, case['mnph']) for _ in range(M)] for _ in range(2)]
    ref_b = engine.Batch(g, B, 1, case['mnph'])
    ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, ref_b.node_capacity, ref_b.edge_capacity, B)
    P0 = PC.flatten_params(ws, PC.make_ref_model(4, 5, seed=4))

    def run_host():
        P, M1, M2 = P0.copy(), np.zeros_like(P0), np.zeros_like(P0)
        G, out, loss = np.zeros_like(P0), np.zeros(B, np.float32), np.zeros(2, np.float32)
        lib.call('igmc_model_set_ctrl', ws.handle, None)
        rec = []
        for t in range(T):
            ref_b.extract(lu, lv, ly, perm, t * B, B, 1.0, 7, 3)
            ref_b.edge_dropout(0.2, False, 7, (3 << 32) ^ t)
            ws.loss_grad(P.ctypes.data, ref_b, out.ctypes.data, G.ctypes.data, loss.ctypes.data, use_edge_flags=True,
                         seed=7, step=11 + t, ARR=0.001)
            ws.adam_step(P.ctypes.data, G.ctypes.data, M1.ctypes.data, M2.ctypes.data, t + 1, 1e-3)
            d = ref_b.download()
            rec.append((d['node_gid'].copy(), d['eflag'].copy(), out.copy(), loss.copy(), P.copy()))
        return rec

    def run_groups(corrupt=False, grouped=False):
        P, M1, M2 = P0.copy(), np.zeros_like(P0), np.zeros_like(P0)
        G, out, loss = np.zeros_like(P0), np.zeros(B, np.float32), np.zeros(2, np.float32)
        total = np.zeros(1, np.float64)
        ctrl = group_ctrl_words(step=11, epoch=3, adam_t=1, batch=B, group=M)
        cp = C.c_void_p(ctrl.ctypes.data)
        for s in sets:
            for a in s:
                lib.call('igmc_batch_set_ctrl', a.handle, cp)
        lib.call('igmc_model_set_ctrl', ws.handle, cp)

        def extract(q, i):
            sets[q][i].extract(lu, lv, ly, perm, q | (i << 1), B, 1.0, 7, 999)        # epoch argument ignored
            sets[q][i].edge_dropout(0.2, False, 7, q | (i << 1))

        bsets = None
        if grouped:      # igmc_extract_group: the whole group (extraction + dropout on the dense blocks) in one launch per stage
            for s in sets:
                for a in s:
                    a.set_lean(True)
            bsets = [engine.BatchSet(s) for s in sets]

        def train(arena):
            lib.call('igmc_train_step', ws.handle, C.c_void_p(P.ctypes.data), arena.handle, 1, None, 7, 0, 1.0, 0.001,
                     C.c_void_p(out.ctypes.data), C.c_void_p(G.ctypes.data), C.c_void_p(M1.ctypes.data),
                     C.c_void_p(M2.ctypes.data), C.c_void_p(loss.ctypes.data), C.c_void_p(total.ctypes.data), cp, 1,
                     1e-3, 0.9, 0.999, 1e-8, 0.0, None)

        if grouped:
            bsets[0].extract(M, lu, lv, ly, perm, 0, B, 1.0, 7, drop_p=0.2, drop_seed=7)
        else:
            for i in range(M):
                extract(0, i)
        rec, gq, t = [], 0, 0
        while t < T:
            steps = min(M, T - t)
            if grouped:
                bsets[1 - gq].extract(M, lu, lv, ly, perm, 1 - gq, B, 1.0, 7, drop_p=0.2, drop_seed=7)
            for i in range(0, M, 2):                    # part of the prefetch before the group's steps ...
                if not grouped:
                    extract(1 - gq, i)
            for i in range(steps):
                if i == 1 and not grouped:
                    for j in range(1, M, 4):            # ... part between them ...
                        extract(1 - gq, j)
                if corrupt and t + i == 4:
                    sets[gq][i].extract(lu, lv, ly, perm, gq | (((i + 1) % M) << 1), B, 1.0, 7, 999)
                train(sets[gq][i])
                d = sets[gq][i].download()
                rec.append((d['node_gid'].copy(), d['eflag'].copy(), out.copy(), loss.copy(), P.copy()))
            for j in range(3, M, 4):                    # ... and the rest after the last tick of the group
                if not grouped:
                    extract(1 - gq, j)
            t += steps
            if steps == M:
                gq ^= 1
        for s in sets:
            for a in s:
                lib.call('igmc_batch_set_ctrl', a.handle, None)
                a.set_lean(False)
        lib.call('igmc_model_set_ctrl', ws.handle, None)
        return rec, ctrl, total

    host = run_host()
    rec, ctrl, total = run_groups()
    K = _lib.CTRL
    assert ctrl[K['SYNC_ERR']] == 0
    assert ctrl[K['STEP']] == 11 + T and ctrl[K['K']] == T and ctrl[K['ADAM_T']] == 1 + T
    assert ctrl[K['GQ']] == (T // M) & 1 and ctrl[K['GK']] == T % M
    assert ctrl[K['FIRST']] == 2 * M * B and ctrl[K['FIRST_ODD']] == 3 * M * B      # each parity finished one group
    assert total[0] == pytest.approx(sum(float(r[3][0]) * B for r in rec), rel=1e-6)
    for a, b in zip(host, rec):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for x, y in zip(a[2:], b[2:]):
            np.testing.assert_allclose(x, y, rtol=2e-5, atol=1e-7)
    _, ctrl_bad, _ = run_groups(corrupt=True)
    assert ctrl_bad[K['SYNC_ERR']] & 2
    # igmc_extract_group (every batch of a group in one launch per extraction stage, edge dropout included) walks the very
    # same trajectory: same node sets, same keep flags, same parameters, bit for bit
    rec_g, ctrl_g, _ = run_groups(grouped=True)
    assert ctrl_g[K['SYNC_ERR']] == 0
    for a, b in zip(rec, rec_g):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


@pytest.mark.parametrize('paths', ['1', '1:4', '0', '0then1', '1then0'])
def test_weight_images_left_by_the_step_equal_the_composed_ones(monkeypatch, paths):
    """igmc_train_step's gradient / Adam kernel also writes the weight images of the parameters it has just updated
    (k_finalize_ts, img), and a caller that asserts unchanged parameters (igmc_model_weights_unchanged) skips k_g2_compose
    at the next call.  Five steps + an evaluation forward with the assertion before every call but the first must give
    exactly what the same calls give when every one of them composes the images from the parameters -- for the subgraph
    kernel's tail (relation-space tables; one workgroup or a cluster of 4 per subgraph), the per-layer kernels' tail
    (basis-space sums), and steps that alternate between the two (images written by one tail, consumed after the other)."""
    be = PC.EmuBackend()
    lib = be.lib
    case = CASES['synth_cap']
    g = engine.Graph(case['A'], lib=lib)
    lu = case['links'][:, 0].astype(np.int32).copy()
    lv = case['links'][:, 1].astype(np.int32).copy()
    ly = case['class_values'][case['link_labels']].astype(np.float32)
    rng = np.random.default_rng(1)
    perm = np.concatenate([rng.permutation(len(lu)) for _ in range(3)]).astype(np.int32)      # (six batches of 4)
    B = 4
    batch = engine.Batch(g, B, 1, case['mnph'])
    ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, batch.node_capacity, batch.edge_capacity, B)
    P0 = PC.flatten_params(ws, PC.make_ref_model(4, 5, seed=4))
    vp = lambda a: C.c_void_p(a.ctypes.data)

    def env_of(i):
        seq = {'0then1': '01010', '1then0': '10101'}.get(paths)
        gs, _, cl = (seq[i] if seq else paths).partition(':')
        monkeypatch.setenv('IGMC_GRAPH_STEP', gs)
        if cl:
            monkeypatch.setenv('IGMC_GS_CLUSTER', cl)

    def run(hint):
        engine.profile_fetch(lib)
        engine.profile_enable(lib, True)
        lib.call('igmc_model_set_ctrl', ws.handle, None)
        P, M1, M2, G = P0.copy(), np.zeros_like(P0), np.zeros_like(P0), np.zeros_like(P0)
        out, loss, total = np.zeros(B, np.float32), np.zeros(2, np.float32), np.zeros(1, np.float64)
        rec = []
        for i in range(5):
            env_of(i)
            batch.extract(lu, lv, ly, perm, i * B, B, 1.0, 7, 3)
            batch.edge_dropout(0.2, False, 7, (3 << 32) ^ i)
            if hint and i > 0:
                lib.call('igmc_model_weights_unchanged', ws.handle, 1)
            lib.call('igmc_train_step', ws.handle, vp(P), batch.handle, 1, None, 7, 11 + i, 1.0, 0.001, vp(out), vp(G),
                     vp(M1), vp(M2), vp(loss), vp(total), None, i + 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, None)
            rec.append((out.copy(), loss.copy(), P.copy(), M1.copy(), M2.copy()))
        env_of(4)
        batch.extract(lu, lv, ly, perm, 5 * B, B, 1.0, 7, 3)
        ev = []
        for rep in range(2):           # evaluation on the trained weights: the second forward also rides on the first's images
            if hint:
                lib.call('igmc_model_weights_unchanged', ws.handle, 1)
            ws.forward(P.ctypes.data, batch, out.ctypes.data, training=False)
            ev.append(out.copy())
        engine.profile_enable(lib, False)
        composes = sum(c for n, _, c in engine.profile_fetch(lib) if n == 'k_g2_compose')
        return rec, ev, composes

    ref, ref_ev, n_ref = run(False)
    uses = {'1': 7, '1:4': 7, '0': 0, '0then1': 2, '1then0': 5}[paths]      # calls of the seven that read the images
    ev_uses = paths in ('1', '1:4', '1then0')                                # ... the two evaluation forwards among them
    assert n_ref == uses              # (the step's tail leaves the images; without the caller's assertion every call composes anyway)
    rec, ev, n = run(True)
    # with the caller's assertion only the very first call composes (not even that one when a step on the per-layer kernels
    # came first: its tail left the images)
    assert n == (1 if ev_uses else 0), n
    for i, (a, b) in enumerate(zip(ref, rec)):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), ('step', i)
    for a, b in zip(ref_ev, ev):
        assert np.array_equal(a, b), 'eval'
    assert np.array_equal(ref_ev[0], ref_ev[1])


def test_host_callback_communicator_reports_a_failing_callback():
    """igmc_comm_create_host: the caller's sum runs inside igmc_allreduce_grads / igmc_train_step_dp; a callback that fails
    (an exception cannot cross the C frame) makes the call fail with the library's error, not silently skip the exchange."""
    from igmc_amd import parallel
    be = PC.EmuBackend()
    lib = be.lib
    seen = []

    def good(ptr, n, stream):
        buf = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
        seen.append(n)
        buf *= 2.0                         # "two ranks holding the same values"

    comm = parallel.HostComm(lib, good, 0, 2)
    assert comm.info() == (0, 2)
    x = np.arange(5, dtype=np.float32)
    lib.call('igmc_allreduce_grads', comm.handle, C.c_void_p(x.ctypes.data), 5, 0.5, None)
    assert seen == [5] and np.array_equal(x, np.arange(5, dtype=np.float32))      # summed, then scaled by 1 / world

    def bad(ptr, n, stream):
        raise ValueError('no peers')

    comm2 = parallel.HostComm(lib, bad, 0, 2)
    with pytest.raises(RuntimeError, match='host all-reduce callback failed'):
        lib.call('igmc_allreduce_grads', comm2.handle, C.c_void_p(x.ctypes.data), 5, 1.0, None)
    with pytest.raises(RuntimeError, match='bad arguments'):
        lib.call('igmc_comm_create_host', None, None, 0, 2, C.byref(C.c_void_p()))
