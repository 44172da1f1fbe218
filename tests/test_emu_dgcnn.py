"""Sort-pool readout family (DGCNN_RS, reference models.py:123-167) -- kernel logic on the CPU emulation vs the oracle."""
import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden

CASES = load_extract_golden()


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


def sub(name, n):
    name, _, cap = name.partition(':')       # 'case:cap' = the case's graph and links with another per-hop cap
    case = dict(CASES[name])
    if cap:
        case['mnph'] = int(cap)
    case['recs'], case['links'], case['link_labels'] = case['recs'][:n], case['links'][:n], case['link_labels'][:n]
    return case


@pytest.mark.parametrize('name,n,R,k,drop', [
    ('synth_nocap', 4, 5, 12, True),       # k smaller than most subgraphs: real selection
    ('synth_cap', 6, 5, 40, False),        # k larger than some subgraphs: zero-padded rows (bias-only conv1 outputs)
    ('hand', 5, 5, 10, True),              # tiny graphs, every one padded
    ('flixster', 4, 10, 14, False),        # 10 relations
])
def test_dgcnn_rs_forward_backward_parity(be, name, n, R, k, drop):
    res = PC.run_dgcnn_parity(be, sub(name, n), R=R, k=k, use_dropout=drop)
    assert res['worst_grad_err'] < 1e-3


def test_sort_pool_order_and_padding(be):
    """global_sort_pool semantics on their own: descending by the last channel, ties in node order, zero rows beyond
    the graph -- the oracle's restatement against a direct numpy statement of the PyG 1.4.2 definition."""
    import torch
    from oracle import pyg_ref
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((11, 5)).astype(np.float32))
    x[3, -1] = x[7, -1]                                   # a tie inside graph 1
    batch = torch.tensor([0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2])
    out = pyg_ref.global_sort_pool(x, batch, 4).view(3, 4, 5).numpy()
    for g in range(3):
        rows = x[batch == g].numpy()
        order = sorted(range(len(rows)), key=lambda i: (-rows[i, -1], i))[:4]
        want = np.zeros((4, 5), np.float32)
        want[:len(order)] = rows[order]
        assert np.array_equal(out[g], want)


def test_sort_pool_error_behaviour(be, monkeypatch):
    """The C ABI refuses what it cannot run and says why (non-zero return + igmc_last_error -> RuntimeError)."""
    from igmc_amd import engine
    case = sub('synth_cap', 4)
    g, b, d = PC.extract_case(be, case, replay=False)
    ws = engine.ModelWorkspace(be.lib, 0, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    slot = b.node_capacity // b.max_graphs
    with pytest.raises(RuntimeError, match='k must be >= 10'):
        engine.SortPoolWorkspace(ws, 9, slot)
    ws_side = engine.ModelWorkspace(be.lib, 0, 5, 4, 4, 6, b.node_capacity, b.edge_capacity, b.max_graphs)
    with pytest.raises(RuntimeError, match='no side features'):
        engine.SortPoolWorkspace(ws_side, 12, slot)
    sp_small = engine.SortPoolWorkspace(ws, 12, max(2, slot // 2))      # created for smaller subgraphs than the arena holds
    P = np.zeros(sp_small.n_params, np.float32)
    out = np.zeros(4, np.float32)
    with pytest.raises(RuntimeError, match='slots larger'):
        sp_small.forward(P.ctypes.data, b, out.ctypes.data)
    sp = engine.SortPoolWorkspace(ws, 12, slot)
    with pytest.raises(RuntimeError, match='null buffer'):
        sp.forward(None, b, out.ctypes.data)


def test_sortpool_kernels_against_the_reference_models_py_fixture(be):
    """``DGCNN_RS`` as the reference's own ``models.py:123-167`` ran it (``tests/golden/model_golden.npz``): k from the
    percentile form, eval outputs, first-step outputs and every gradient on the recorded subgraphs / masks."""
    from helpers import load_model_golden
    res = PC.run_reference_fixture_dgcnn(be, load_model_golden('dgcnn_rs'), 8)
    assert res['k'] == 40


@pytest.mark.parametrize('form', ['tables', 'per_layer'])
@pytest.mark.parametrize('name,n,k,drop', [('synth_cap', 6, 40, True), ('synth_nocap:100', 4, 30, False)])
def test_dgcnn_rs_on_the_dense_layer_kernels(be, monkeypatch, name, n, k, drop, form):
    """The conv layers of the sort-pool family on the dense-layer kernels (arenas with the transposed block): forward as ONE
    launch, backward either as ONE launch with relation-space tables -- dPre_3 of every row and the per-row readout gradient
    ``dcat`` instead of the centre-node head's target rows -- or one launch per layer pass (G / Y form, IGMC_DL_TS=0)."""
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    if form == 'per_layer':
        monkeypatch.setenv('IGMC_DL_TS', '0')
    from igmc_amd import engine
    engine.profile_enable(be.lib, True)
    try:
        res = PC.run_dgcnn_parity(be, sub(name, n), R=5, k=k, use_dropout=drop)
        ran = [nm for nm, _, _ in engine.profile_fetch(be.lib)]
    finally:
        engine.profile_enable(be.lib, False)
    assert res['worst_grad_err'] < 1e-3
    assert 'k_dl_fwd' in ran
    assert ('k_dl_bwd' in ran) == (form == 'tables') and ('k_dl_layer_bwd' in ran) == (form == 'per_layer'), ran
