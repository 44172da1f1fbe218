"""Kernel LOGIC of the HIP model kernels (igmc_amd/csrc/model.hip: basis-space gather, f32-MFMA dense
transforms, layer-0 tables, head, backward, ARR, Adam) on the CPU emulation of the same sources, checked
against the PyG-1.4.2 restatement (oracle/pyg_ref.py) on identical subgraphs / weights / dropout masks."""
import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden

CASES = load_extract_golden()


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


@pytest.fixture(autouse=True)
def _graph_step_path(monkeypatch):
    # capped cases go through the opt-in one-workgroup-per-subgraph kernel (graphstep.hip); uncapped ones are not
    # eligible for it and keep exercising the per-layer kernels
    monkeypatch.setenv('IGMC_GRAPH_STEP', '1')


def sub(name, n):
    name, _, cap = name.partition(':')       # 'case:cap' = the case's graph and links with another per-hop cap
    case = dict(CASES[name])
    if cap:
        case['mnph'] = int(cap)
    case['recs'], case['links'], case['link_labels'] = case['recs'][:n], case['links'][:n], case['link_labels'][:n]
    return case


@pytest.mark.parametrize('name,n,R,drop,mult', [
    ('synth_nocap', 3, 5, True, 1.0),      # ML-like, edge dropout + MLP dropout masks injected
    ('flixster', 5, 10, False, 1.0),       # 10 relations
    ('yahoo_music', 4, 71, True, 20.0),    # 71 relations (shared layer-0 table path), multiply_by
    ('hand_h2', 5, 5, True, 1.0),          # 2 hops -> 6 node labels
    # capped subgraphs: the one-workgroup-per-subgraph kernel (graphstep.hip)
    ('synth_cap', 6, 5, True, 1.0),
    ('douban_cap20', 5, 5, False, 2.0),
    ('hand', 5, 5, True, 1.0),
    ('synth_nocap:45', 3, 5, True, 1.0),   # up to 92 nodes: two 64-row passes per layer
    ('synth_nocap:100', 4, 5, True, 1.0),  # up to 202 nodes (the ml_1m shape): 13 bundles, ranked schedule
    ('douban:100', 6, 5, False, 1.0),
])
def test_forward_backward_parity(be, name, n, R, drop, mult):
    res = PC.run_model_parity(be, sub(name, n), R=R, use_dropout=drop, multiply_by=mult)
    assert res['worst_grad_err'] < 1e-4


@pytest.mark.parametrize('cs', ['2', '4'])
@pytest.mark.parametrize('name,n,drop', [('synth_cap', 6, True), ('synth_nocap:100', 4, False), ('hand', 5, True)])
def test_graph_step_clusters(be, monkeypatch, cs, name, n, drop):
    """Workgroup clusters of the subgraph kernel (2 / 4 workgroups share a subgraph and exchange h_l / dPre_l rows as
    tagged words): the emulator runs the members of a cluster together (hipemu::Runtime::co_cs), so the bundle
    schedule over 8 / 16 waves, the exchange indices and tags, the per-member partial tables and their reduction are
    checked on the CPU against the oracle like the single-workgroup path."""
    monkeypatch.setenv('IGMC_GS_CLUSTER', cs)
    res = PC.run_model_parity(be, sub(name, n), R=5, use_dropout=drop)
    assert res['worst_grad_err'] < 1e-4


def test_hand_off_finalize_variant(be, monkeypatch):
    # IGMC_FIN_MODE=0: subgraph kernel + k_finalize (in-kernel hand-offs) instead of the default k_finalize_ts
    monkeypatch.setenv('IGMC_FIN_MODE', '0')
    res = PC.run_model_parity(be, sub('synth_cap', 6), R=5, use_dropout=True)
    assert res['worst_grad_err'] < 1e-4


@pytest.mark.parametrize('n_side', [32, 10])
def test_side_features(be, n_side):
    """--use-features path (reference models.py:186-188,208-209): lin1 widens by n_side; 32 -> MFMA head
    kernels (D %% 16 == 0), 10 -> the generic head kernels."""
    res = PC.run_model_parity(be, sub('synth_nocap', 3), R=5, use_dropout=False, n_side=n_side)
    assert res['worst_grad_err'] < 1e-4


def test_adam_matches_torch(be):
    import torch
    rng = np.random.default_rng(0)
    n = 5000
    p0 = rng.standard_normal(n).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-3, weight_decay=0.01)
    P, M1, M2 = be.dev(p0.copy()), be.dev(np.zeros(n, np.float32)), be.dev(np.zeros(n, np.float32))
    from igmc_amd import engine

    class W(object):
        pass
    for step in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        G = be.dev(g)
        be.lib.call('igmc_adam_step', engine._p(be.ptr(P)), engine._p(be.ptr(G)), engine._p(be.ptr(M1)),
                    engine._p(be.ptr(M2)), n, step, 1e-3, 0.9, 0.999, 1e-8, 0.01, None)
    np.testing.assert_allclose(be.host(P), tp.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_train_steps_follow_reference_trajectory(be):
    """3 optimiser steps with injected masks: parameters track torch Adam on the oracle gradients."""
    import torch
    from oracle import pyg_ref
    from helpers import batch_to_pyg
    res = PC.run_model_parity(be, sub('synth_nocap', 3), R=5, use_dropout=False, check_eval=False)
    ws, b, ref, d = res['ws'], res['batch'], res['ref'], res['d']
    P = be.dev(res['flat'].copy())
    M1, M2 = be.dev(np.zeros(ws.n_params, np.float32)), be.dev(np.zeros(ws.n_params, np.float32))
    G, out, loss = be.dev(np.zeros(ws.n_params, np.float32)), be.dev(np.zeros(d['B'], np.float32)), be.dev(np.zeros(2, np.float32))
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    pyg = batch_to_pyg(d, 4)
    rng = np.random.default_rng(9)
    for step in range(1, 4):
        lm = rng.random((d['B'], 128)) < 0.5
        LM = be.dev(lm.astype(np.uint8).reshape(-1))
        ws.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(G), be.ptr(loss), lin_mask=be.ptr(LM), ARR=0.001)
        ws.adam_step(be.ptr(P), be.ptr(G), be.ptr(M1), be.ptr(M2), step, 1e-3)
        ref_loss = pyg_ref.train_step(ref, opt, pyg, ARR=0.001, lin_mask=torch.from_numpy(lm))
        assert be.host(loss)[0] == pytest.approx(ref_loss, rel=5e-4)
    np.testing.assert_allclose(be.host(P), PC.flatten_params(ws, ref), rtol=2e-3, atol=2e-5)


@pytest.mark.parametrize('name,n', [('synth_cap', 8), ('synth_nocap', 3)])
@pytest.mark.parametrize('force_undirected', [False, True])
def test_free_running_dropout(be, name, n, force_undirected):
    """k_edge_flags / the MLP-dropout hash drawn by the kernels themselves (capped: subgraph kernel; uncapped: per-layer
    kernels): mask statistics, bit-exactness vs the host restatement of include/igmc_rng.h, model parity with the drawn
    masks incl. ``force_undirected`` through the oracle's own ``dropout_adj(force_undirected=True)``."""
    res = PC.run_free_running_dropout(be, sub(name, n), R=5, p=0.2, force_undirected=force_undirected)
    assert res['worst_grad_err'] < 1e-4


@pytest.mark.parametrize('name,cluster,drop', [('synth_cap', '1', True), ('synth_cap', '4', False), ('synth_nocap', '1', True),
                                               ('flixster', '1', True), ('yahoo_music', '1', True)])
def test_fused_train_step_tracks_torch_adam(be, monkeypatch, name, cluster, drop):
    """``igmc_train_step`` (k_graph_step -> k_tail_ts -> k_finalize_ts incl. Adam for the capped case; per-layer kernels +
    k_finalize_ts in basis-space mode for the uncapped one and for flixster's 10 relations, whose layer-0 table comes from
    k_l0_bwd; yahoo_music's 71 relations: k_finalize's hand-off tail) over 4 different batches vs
    ``pyg_ref.train_step`` + ``torch.optim.Adam``."""
    monkeypatch.setenv('IGMC_GS_CLUSTER', cluster)
    R = {'flixster': 10, 'yahoo_music': 71}.get(name, 5)
    n = 8 if name == 'yahoo_music' else 16
    res = PC.run_fused_train_trajectory(be, sub(name, n), R=R, steps=4, batch=n // 4, use_dropout=drop)
    assert res['frac_off'] < 2e-3


def test_side_source_gathered_by_the_extraction_launch(be):
    """``igmc_batch_bind_side_source``: the target nodes' feature rows (reference util_functions.py:250-253) are
    gathered on the device by the extraction launch through the batch's own (permutation, first) indexing -- the same
    forward as handing the gathered [B, S] rows over with ``igmc_batch_set_side_features``."""
    from igmc_amd import engine
    case = sub('synth_nocap', 8)
    A, S, B = case['A'], 32, 4
    g = engine.Graph(A, lib=be.lib)
    n = len(case['links'])
    lu, lv = case['links'][:, 0].astype(np.int32).copy(), case['links'][:, 1].astype(np.int32).copy()
    ly = case['class_values'][case['link_labels']].astype(np.float32)
    side_all = np.random.default_rng(3).standard_normal((n, S)).astype(np.float32)
    perm = np.array([5, 2, 7, 0, 3, 1, 6, 4], np.int32)
    ref = PC.make_ref_model(4, 5, n_side=S, seed=2)
    outs = []
    for mode in ('bound', 'handed'):
        b = engine.Batch(g, B, 1, case['mnph'])
        ws = engine.ModelWorkspace(be.lib, 0, 5, 4, 4, S, b.node_capacity, b.edge_capacity, B)
        P = PC.flatten_params(ws, ref)
        if mode == 'bound':
            b.bind_side_source(side_all.ctypes.data, S)
        b.extract(lu.ctypes.data, lv.ctypes.data, ly.ctypes.data, perm.ctypes.data, 2, B)
        if mode == 'handed':
            rows = np.ascontiguousarray(side_all[perm[2:2 + B]])
            b.set_side_features(rows.ctypes.data, S)
        out = np.zeros(B, np.float32)
        ws.forward(P.ctypes.data, b, out.ctypes.data, training=False)
        outs.append(out.copy())
    assert np.array_equal(outs[0], outs[1]) and np.all(np.isfinite(outs[0])) and np.ptp(outs[0]) > 0


def test_lean_extraction_feeds_the_dense_kernel(be):
    """A lean arena (``igmc_batch_set_lean``) skips the CSR emission; the matrix-core subgraph kernel must give the same
    loss / gradients from the dense blocks alone, and the CSR appears on demand (download) identical to the eager one."""
    from igmc_amd import engine
    case = sub('synth_cap', 8)
    A = case['A']
    g = engine.Graph(A, lib=be.lib)
    lu, lv = case['links'][:, 0].astype(np.int32).copy(), case['links'][:, 1].astype(np.int32).copy()
    ly = case['class_values'][case['link_labels']].astype(np.float32)
    ref = PC.make_ref_model(4, 5, seed=6)
    res = {}
    for mode in ('eager', 'lean'):
        b = engine.Batch(g, 8, 1, case['mnph'])
        ws = engine.ModelWorkspace(be.lib, 0, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, 8)
        assert ws.dense_path(b, 8)
        b.set_lean(mode == 'lean')
        b.extract(lu.ctypes.data, lv.ctypes.data, ly.ctypes.data, None, 0, 8, seed=3, epoch=1)
        P = PC.flatten_params(ws, ref)
        out, grad, loss = np.zeros(8, np.float32), np.zeros(ws.n_params, np.float32), np.zeros(2, np.float32)
        lm = (np.random.default_rng(1).random((8, 128)) < 0.5).astype(np.uint8)
        ws.loss_grad(P.ctypes.data, b, out.ctypes.data, grad.ctypes.data, loss.ctypes.data, lin_mask=lm.ctypes.data, ARR=0.001)
        d = b.download()
        res[mode] = (out.copy(), grad.copy(), loss.copy(), d)
    assert np.array_equal(res['eager'][0], res['lean'][0]) and np.array_equal(res['eager'][1], res['lean'][1])
    assert np.array_equal(res['eager'][2], res['lean'][2])
    for k in ('node_off', 'row_ptr', 'col', 'erel', 'node_label', 'node_gid', 'y'):
        assert np.array_equal(res['eager'][3][k], res['lean'][3][k]), k
    PC.check_batch_structure(res['lean'][3], 4)


def test_uncapped_sparse_graph_takes_the_dense_kernel(be, monkeypatch):
    """No per-hop cap (the reference default for the Monti datasets): the slot capacity of a hop-1 arena is bounded by
    the longest row / column of the rating matrix, so a sparse graph (douban: at most 111 raters per item, 108 items per
    user) still gets dense induced blocks and the matrix-core subgraph kernel -- checked against the oracle with
    injected edge-dropout and MLP-dropout masks like every other path."""
    import scipy.sparse as sp
    from igmc_amd import engine
    monkeypatch.setenv('IGMC_GS_CLUSTER', '4')      # what an MI355X picks for batches of <= 56 links (4 x 32 rows a side)
    case = sub('douban', 6)
    assert case['mnph'] is None or case['mnph'] < 0 or case['mnph'] >= 10000
    A = sp.csr_matrix(case['A'])
    A.eliminate_zeros()
    max_row = int(np.diff(A.indptr).max())
    max_col = int(np.diff(A.tocsc().indptr).max())
    assert max(max_row, max_col) < 128
    g = engine.Graph(case['A'], lib=be.lib)
    b = engine.Batch(g, 6, 1, case['mnph'])
    ws = engine.ModelWorkspace(be.lib, 0, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, 6)
    assert b.node_capacity == 6 * ((1 + max_col) + (1 + max_row))
    assert ws.dense_path(b, 6)
    res = PC.run_model_parity(be, case, R=5, use_dropout=True)
    assert res['worst_grad_err'] < 1e-4
    assert res['ws'].dense_path(res['batch'], 6)


@pytest.mark.parametrize('force_undirected', [False, True])
def test_free_running_dropout_on_a_lean_arena(be, force_undirected):
    """Edge dropout of a lean arena is drawn on the dense blocks (k_relm_dropout, no CSR): the flags of the CSR emitted
    afterwards must be the same counter-based draws as k_edge_flags makes (host restatement, bit for bit), and the
    subgraph kernel's loss / gradients with them must match the oracle."""
    res = PC.run_free_running_dropout(be, sub('synth_cap', 12), R=5, force_undirected=force_undirected, lean=True)
    assert res['worst_grad_err'] < 1e-4
    eager = PC.run_free_running_dropout(be, sub('synth_cap', 12), R=5, force_undirected=force_undirected, lean=False)
    assert eager['keep_rate'] == res['keep_rate']


@pytest.mark.parametrize('name,n,drop', [('synth_cap', 6, True), ('synth_nocap:100', 4, True), ('synth_nocap:100', 4, False),
                                         ('hand', 5, True), ('douban:100', 6, False),
                                         ('synth_nocap:200', 4, True)])      # > 128 rows a side: two workgroups per side
@pytest.mark.parametrize('lean', [False, True])
def test_dense_per_layer_kernels(be, monkeypatch, name, n, drop, lean):
    """k_dl_layer (graphstep2.hip): the conv layers of the per-layer sequence on the matrix cores, for arenas whose slots
    are too large for the subgraph kernel (ml_100k, cap 200) -- here forced onto small cases (IGMC_DL_ALWAYS allocates the
    transposed block, IGMC_GRAPH_STEP=0 keeps the subgraph kernels away): forward, loss and every gradient vs the oracle."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    # (lean: the arena stops at the dense blocks; the model calls emit the node arrays only -- layer 0 included, nothing in
    #  the step reads an edge list)
    res = PC.run_model_parity(be, sub(name, n), R=5, use_dropout=drop, lean=lean)
    assert res['worst_grad_err'] < 1e-4
    assert res['batch'].dense_layers(res['ws'])
    if name.endswith(':200'):
        d = res['d']
        sizes = np.diff(np.asarray(d['node_off']))
        nu = np.asarray(d['n_users'])
        assert max(nu.max(), (sizes - nu).max()) > 128, 'this case is meant to need a second workgroup per side'


def test_dense_per_layer_kernels_in_the_fused_train_step(be, monkeypatch):
    """... and inside igmc_train_step (the per-layer sequence with the multi-role launches + fused Adam): five steps on
    different batches track pyg_ref.train_step + torch.optim.Adam."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    res = PC.run_fused_train_trajectory(be, sub('synth_cap', 15), R=5, steps=5, batch=3, use_dropout=True)
    assert res['frac_off'] < 2e-3


def test_dense_layers_launch_forms_agree_bit_for_bit(be, monkeypatch):
    """k_dl_fwd / k_dl_bwd (one launch per direction, members of a subgraph exchanging rows), the forward alone as one
    launch, one launch per layer pass: the same arithmetic in the same order, bit-identical parameters after the steps.
    The case has two workgroups per side (more than 128 rows)."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    runs = {}
    for fused in ('2', '1', '0'):
        monkeypatch.setenv('IGMC_DL_FUSED', fused)
        runs[fused] = PC.run_fused_train_trajectory(be, sub('synth_nocap:200', 8), R=5, steps=2, batch=4, use_dropout=True)
        assert runs[fused]['frac_off'] < 2e-3
    for other in ('1', '0'):
        for k in ('params', 'm1', 'm2'):
            assert np.array_equal(runs['2'][k], runs[other][k]), (other, k)
    monkeypatch.setenv('IGMC_DL_TS', '0')          # the G / Y form of the backward: another order of the sums
    gy = PC.run_fused_train_trajectory(be, sub('synth_nocap:200', 8), R=5, steps=2, batch=4, use_dropout=True)
    assert gy['frac_off'] < 2e-3 and not np.array_equal(gy['params'], runs['2']['params'])


def test_dense_per_layer_kernels_with_side_features(be, monkeypatch):
    """The dense per-layer kernels only replace the conv layers: a model with side features (centre-node readout + the
    two targets' feature rows, reference models.py:208-209) takes them too."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    res = PC.run_model_parity(be, sub('synth_cap', 6), R=5, use_dropout=True, n_side=10)
    assert res['worst_grad_err'] < 1e-4
    assert res['batch'].dense_layers(res['ws'])
    # a width the one-launch-per-direction sequence takes (D = 256 + 16): k_dl_fwd -> k_head_sub with the side columns in its
    # lin1 rows -> k_dl_bwd -> the table tail
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    monkeypatch.setenv('IGMC_GS_TRACE', '1')
    res = PC.run_model_parity(be, sub('synth_nocap:200', 4), R=5, use_dropout=True, n_side=16)
    assert res['worst_grad_err'] < 1e-4
    assert res['batch'].dense_layers(res['ws'])


@pytest.mark.parametrize('case', ['igmc_r5', 'igmc_side', 'igmc_r10'])
def test_kernels_against_the_reference_models_py_fixtures(be, case):
    """The kernel logic against outputs of the reference's OWN ``models.py`` / ``train_eval.py``
    (``tests/golden/model_golden.npz``, ``make_model_golden.py``): same subgraphs (recorded node lists replayed), weights,
    edge / MLP dropout masks; eval outputs, first-step outputs + every gradient, then the epoch through the fused step
    against the parameters the reference's ``train`` + Adam left and its returned epoch loss."""
    from helpers import load_model_golden
    res = PC.run_reference_fixture(be, load_model_golden(case), 8)
    assert res['worst_grad_rel'] < PC.GRAD_TOL and res['params_frac_off'] == 0.0


@pytest.mark.parametrize('drop,lean', [(False, False), (True, True)])
def test_dense_layers_with_ten_relations(be, monkeypatch, drop, lean):
    """More than five relations on the matrix cores (flixster: ten rating levels, reference Main.py:387
    num_relations = len(class_values)): k_dl_fwd / k_dl_bwd take the relations in groups of five -- one weight image per group,
    a 64-row layer-0 table -- and the relation-space tables' tail turns them into gradients.  Forward, loss and every
    gradient vs the oracle."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    res = PC.run_model_parity(be, sub('flixster', 5), R=10, use_dropout=drop, lean=lean)
    assert res['worst_grad_err'] < 1e-4
    assert res['batch'].dense_layers(res['ws'])


def test_ten_relations_in_the_fused_train_step(be, monkeypatch):
    """... and inside igmc_train_step: four steps on different batches track pyg_ref.train_step + torch.optim.Adam."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GS_TRACE', '1')
    res = PC.run_fused_train_trajectory(be, sub('flixster', 16), R=10, steps=4, batch=4, use_dropout=True)
    assert res['frac_off'] < 2e-3


def test_free_running_dropout_on_a_lean_arena_with_ten_relations(be, monkeypatch):
    """... with relation codes past 7 in the block bytes (flixster): every edge takes part in the draws on the dense blocks
    (the same flags as the eager CSR draws), and the relation-group kernels with them match the oracle."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    res = PC.run_free_running_dropout(be, sub('flixster', 16), R=10, force_undirected=False, lean=True)
    assert res['worst_grad_err'] < 1e-4
    eager = PC.run_free_running_dropout(be, sub('flixster', 16), R=10, force_undirected=False, lean=False)
    assert eager['keep_rate'] == res['keep_rate']


def ten_levels(case):
    """An ML-like case with every rating level split into two half-star levels (by a hash of the pair): ten relations
    on the case's own graph shape -- ml_10m's levels on an ml_1m-like graph (bench.py's ``ml_10m_lite``)."""
    c = dict(case)
    A = case['A'].tocoo()
    half = ((A.row.astype(np.int64) * 2654435761 + A.col.astype(np.int64) * 40503) >> 7) & 1
    import scipy.sparse as sp
    c['A'] = sp.csr_matrix((2.0 * A.data - 1.0 + half, (A.row, A.col)), shape=A.shape)
    c['class_values'] = np.arange(1, 11, dtype=np.float64) / 2.0
    lab = np.asarray(case['link_labels'], np.int64)
    c['link_labels'] = 2 * lab + (np.arange(len(lab)) & 1)
    c['recs'] = [None] * len(case['links'])
    return c


@pytest.mark.parametrize('drop,lean', [(False, True), (True, False)])
def test_ten_relations_at_cap_100(be, drop, lean):
    """R = 10 x cap 100 (up to 101 rows a side: two workgroups per side AND two relation groups in one launch) -- the
    ml_10m_lite shape of bench.py, GPU twin: test_gpu_headline.py::test_ml10m_lite_batch_matches_oracle."""
    case = ten_levels(sub('synth_nocap:100', 4))
    res = PC.run_model_parity(be, case, R=10, use_dropout=drop, lean=lean)
    assert res['worst_grad_err'] < 1e-4
    assert res['batch'].dense_layers(res['ws'])
    assert int(res['d']['erel'].max()) >= 8


@pytest.mark.parametrize('drop', [False, True])
def test_group_split_equals_group_after_group(be, monkeypatch, drop):
    """Ten relations with at most four bundles a workgroup (cap 100; flixster): waves 0..3 take the bundles' first relation
    group and waves 4..7 the second one at the same time (k_dl_fwd / k_dl_bwd<..., GS>), instead of every wave taking the two
    groups one after the other (IGMC_DL_GSPLIT=0).  Same sums in the same order: outputs, loss, every gradient bit for bit."""
    res = {}
    for tag in ('split', 'serial'):
        if tag == 'serial':
            monkeypatch.setenv('IGMC_DL_GSPLIT', '0')
        r = PC.run_model_parity(be, ten_levels(sub('synth_nocap:100', 4)), R=10, use_dropout=drop)
        assert r['worst_grad_err'] < 1e-4 and r['batch'].dense_layers(r['ws'])
        res[tag] = r
    for k in ('train_out', 'loss'):
        assert np.array_equal(np.asarray(res['split'][k]), np.asarray(res['serial'][k])), k
    assert set(res['split']['grads']) == set(res['serial']['grads']) and res['split']['grads']
    for k, g in res['split']['grads'].items():
        assert np.array_equal(g, res['serial']['grads'][k]), k


def test_ten_relations_at_cap_100_in_the_fused_train_step(be):
    case = ten_levels(sub('synth_nocap:100', 8))
    runs = [PC.run_fused_train_trajectory(be, case, R=10, steps=4, batch=2, use_dropout=True) for _ in range(2)]
    assert runs[0]['frac_off'] < PC.TRAJ_FRAC_OFF
    for k in ('params', 'm1', 'm2'):
        assert np.array_equal(runs[0][k], runs[1][k]), k


@pytest.mark.parametrize('name,R,n', [('hand_h2', 5, 5), ('flixster_h2', 10, 4)])
def test_two_hops_take_the_dense_layers(be, name, R, n):
    """Two hops (six node labels, reference util_functions.py:248-262): the layer-0 table [R L | L | 1] has 37 (R = 5) rows --
    more than the 32 of the one-group layout -- so the model takes the two-group layout of the dense-layer kernels (64-row
    table; with five relations the second group holds none and is skipped); flixster's ten relations x six labels = 67 rows
    stay on the per-layer kernels."""
    res = PC.run_model_parity(be, sub(name, n), R=R, use_dropout=True)
    assert res['worst_grad_err'] < 1e-4
    assert bool(res['batch'].dense_layers(res['ws'])) == (R == 5)
