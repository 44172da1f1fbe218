"""DGCNN_RS (reference models.py:123-167; the sort-pool readout family) on the MI355X: kernels vs the oracle through the
C ABI, and the reference-shaped Python surface (constructor, state_dict, train / eval through train_eval)."""
import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden

pytestmark = pytest.mark.gpu
CASES = load_extract_golden()


@pytest.fixture(scope='module')
def be():
    return PC.GpuBackend()


def sub(name, n):
    name, _, cap = name.partition(':')
    case = dict(CASES[name])
    if cap:
        case['mnph'] = int(cap)
    case['recs'], case['links'], case['link_labels'] = case['recs'][:n], case['links'][:n], case['link_labels'][:n]
    return case


@pytest.mark.parametrize('name,n,R,k,drop', [
    ('synth_nocap', 16, 5, 12, True),
    ('synth_nocap:100', 16, 5, 60, True),      # the ml_1m shape (up to 202 nodes), k = 60
    ('synth_cap', 16, 5, 40, False),           # k beyond some subgraphs: zero-padded rows
    ('douban', 24, 5, 30, True),               # the reference's k = 30 default on real subgraphs
    ('flixster', 48, 10, 14, False),
    ('hand', 5, 5, 10, True),
])
def test_dgcnn_rs_forward_backward_parity(be, name, n, R, k, drop):
    res = PC.run_dgcnn_parity(be, sub(name, n), R=R, k=k, use_dropout=drop)
    assert res['worst_grad_err'] < 2e-3


@pytest.mark.parametrize('form', ['tables', 'per_layer'])
@pytest.mark.parametrize('name,n,k,drop', [('synth_nocap:100', 16, 60, True), ('douban', 24, 30, True)])
def test_dgcnn_rs_on_the_dense_layer_kernels(be, monkeypatch, name, n, k, drop, form):
    """The conv layers of the sort-pool family on the dense-layer kernels (what the step graph's arenas take): forward as ONE
    launch; backward as ONE launch with relation-space tables -- dPre_3 of every row, the per-row readout gradient -- or one
    launch per layer pass (IGMC_DL_TS=0).  Loss and every gradient vs the oracle."""
    from igmc_amd import engine
    monkeypatch.setenv('IGMC_DL_ALWAYS', '1')
    monkeypatch.setenv('IGMC_GRAPH_STEP', '0')
    if form == 'per_layer':
        monkeypatch.setenv('IGMC_DL_TS', '0')
    engine.profile_enable(be.lib, True)
    try:
        res = PC.run_dgcnn_parity(be, sub(name, n), R=5, k=k, use_dropout=drop)
        ran = [nm for nm, _, _ in engine.profile_fetch(be.lib)]
    finally:
        engine.profile_enable(be.lib, False)
    assert res['worst_grad_err'] < 2e-3
    assert 'k_dl_fwd' in ran
    assert ('k_dl_bwd' in ran) == (form == 'tables') and ('k_dl_layer_bwd' in ran) == (form == 'per_layer'), ran


def test_dgcnn_rs_bitwise_reproducible(be):
    r1 = PC.run_dgcnn_parity(be, sub('synth_nocap', 16), R=5, k=20)
    r2 = PC.run_dgcnn_parity(be, sub('synth_nocap', 16), R=5, k=20)
    assert np.array_equal(r1['eval_out'], r2['eval_out']) and np.array_equal(r1['loss'], r2['loss'])


def test_dgcnn_rs_python_surface_trains():
    """Constructor arguments / attributes / state_dict keys of the reference class, the percentile form of k
    (models.py:70-73), and a few epochs through train_multiple_epochs: the loss falls and evaluation agrees with the
    oracle's DGCNN_RS on the trained weights."""
    import math
    import torch
    from igmc_amd import preprocessing
    from igmc_amd.models import DGCNN_RS
    from igmc_amd.train_eval import DataLoader, FlatAdam, eval_rmse, train
    from igmc_amd.util_functions import MyDataset, MyDynamicDataset
    from oracle import pyg_ref
    split = preprocessing.load_data_monti('douban', testing=True)
    (_, _, adj, trl, tru, trv, _, _, _, tel, teu, tev, cv) = split
    tr = MyDynamicDataset('data/t/dg_train', adj, (tru[:800], trv[:800]), trl[:800], 1, 1.0, 10000, None, None, cv)
    te = MyDataset('data/t/dg_test', adj, (teu[:200], tev[:200]), tel[:200], 1, 1.0, 10000, None, None, cv)
    torch.manual_seed(3)
    model = DGCNN_RS(tr, latent_dim=[32, 32, 32, 1], k=0.6, num_relations=len(cv), num_bases=4, regression=True,
                     adj_dropout=0.2, force_undirected=False).to('cuda')
    sizes = sorted(DGCNN_RS._subgraph_sizes(tr))
    assert model.k == max(10, sizes[int(math.ceil(0.6 * len(sizes))) - 1])
    assert model.dense_dim == (model.k // 2 - 4) * 32 and model.total_latent_dim == 97
    want = {'convs.%d.%s' % (l, p) for l in range(4) for p in ('basis', 'att', 'root', 'bias')} | {
        'conv1d_params1.weight', 'conv1d_params1.bias', 'conv1d_params2.weight', 'conv1d_params2.bias',
        'lin1.weight', 'lin1.bias', 'lin2.weight', 'lin2.bias'}
    sd = model.state_dict()
    assert set(sd) == want
    assert tuple(sd['convs.3.basis'].shape) == (4, 32, 1) and tuple(sd['conv1d_params1.weight'].shape) == (16, 1, 97)
    assert tuple(sd['conv1d_params2.weight'].shape) == (32, 16, 5) and tuple(sd['lin1.weight'].shape) == (128, model.dense_dim)
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    loader, tl = DataLoader(tr, 50, shuffle=True), DataLoader(te, 50, shuffle=False)
    losses = [train(model, opt, loader, 'cuda', regression=True, ARR=0.001) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    rmse = eval_rmse(model, tl, 'cuda')
    # the oracle on the same weights and the same (epoch-independent) test subgraphs
    ref = pyg_ref.DGCNNRSRef(4, (32, 32, 32, 1), model.k, len(cv), 4, adj_dropout=0.2, fast=True)
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    ref.eval()
    sse = 0.0

    class PB(object):
        pass
    for data in tl:
        raw = data._materialise()
        pb = PB()
        pb.x, pb.edge_index, pb.edge_type, pb.batch = raw['x'], raw['edge_index'], raw['edge_type'], raw['batch']
        with torch.no_grad():
            o = ref(pb)
        sse += float(((o - data.y.cpu().view(-1)) ** 2).sum())
    assert abs(rmse - math.sqrt(sse / len(te))) < 1e-4


def test_dgcnn_rs_steps_replay_from_the_step_graph():
    """The sort-pool family trains through the same grouped step graph as IGMC (conv kernels + sort-pool forward / backward ->
    igmc_sortpool_step_finish: Adam, loss, control-block tick): groups of 4 replayed from the hipGraph == the same launches
    made eagerly on one stream, bit for bit (parameters, Adam moments, epoch totals), over two epochs with edge dropout."""
    import torch
    from igmc_amd import preprocessing
    from igmc_amd.models import DGCNN_RS
    from igmc_amd.stepgraph import StepGraph
    from igmc_amd.train_eval import FlatAdam
    from igmc_amd.util_functions import MyDynamicDataset
    (_, _, adj, trl, tru, trv, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('douban', testing=True)
    tr = MyDynamicDataset('data/t/dg_graph', adj, (tru[:1000], trv[:1000]), trl[:1000], 1, 1.0, 10000, None, None, cv, seed=2)
    perm = torch.randperm(len(tr), generator=torch.Generator().manual_seed(5))
    res = {}
    for name, kw in (('graph', dict(group=4)), ('eager', dict(group=4, use_graph=False, overlap=False))):
        torch.manual_seed(3)
        model = DGCNN_RS(tr, latent_dim=[32, 32, 32, 1], k=30, num_relations=len(cv), num_bases=4, regression=True,
                         adj_dropout=0.2, seed=1).to('cuda')
        model.reset_parameters()
        opt = FlatAdam(model, lr=1e-3)
        sg = StepGraph(model, opt, tr, 50, 0.001, **kw)
        assert sg.sp is not None
        totals = [float(sg.run_epoch(perm, ep)[0].item()) for ep in (1, 2)]
        torch.cuda.synchronize()
        assert (sg.graph is not None) == (name == 'graph') and opt.t == 40
        res[name] = (model.flat_parameters().detach().cpu().clone(), opt.exp_avg.detach().cpu().clone(),
                     opt.exp_avg_sq.detach().cpu().clone(), totals)
    assert all(np.isfinite(res['graph'][3])) and res['graph'][3][1] < res['graph'][3][0]
    for a, b in zip(res['graph'][:3], res['eager'][:3]):
        assert torch.equal(a, b)
    assert res['graph'][3] == res['eager'][3]


def test_dgcnn_rs_matches_the_reference_fixture(be):
    """The sort-pool family against the reference's own ``DGCNN_RS`` (``models.py:123-167``) run recorded in
    ``tests/golden/model_golden.npz``."""
    from helpers import load_model_golden
    res = PC.run_reference_fixture_dgcnn(be, load_model_golden('dgcnn_rs'), 8)
    assert res['k'] == 40
