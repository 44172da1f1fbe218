"""GPU parity at the REAL headline shape (BASELINE.json configs[2]: ml_1m-shaped rating graph 6040 x 3706, hop 1,
max-nodes-per-hop 100, batch 50 -> k_graph_step with clusters of 4 workgroups per subgraph) against the CPU oracle,
plus the paths round 1 only checked transitively:

* one full batch of 50: extraction (candidate sets, sizes, induced edges vs ``oracle/extract_ref``), eval output,
  train output / loss / EVERY gradient vs ``oracle/pyg_ref`` (tolerances of ``parity_checks.run_model_parity``:
  outputs 2e-5 and gradients 5e-5 of the tensor's peak, loss 3e-6 -- 10x the worst the GPU shows, profiles/r03_parity_observed.txt);
* >= 5 consecutive steps of the fused ``igmc_train_step`` (k_graph_step -> k_tail_ts -> k_finalize_ts incl. Adam) vs
  ``pyg_ref.train_step`` + ``torch.optim.Adam`` (reference train_eval.py:158-177);
* the free-running counter-based dropout draws (edge dropout incl. ``force_undirected``, MLP dropout): statistics,
  bit-exactness vs the host restatement of the hashes, model parity with the drawn masks (reference
  models.py:193-198, :212).
"""
import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden

pytestmark = pytest.mark.gpu
CASES = load_extract_golden()


@pytest.fixture(scope='module')
def be():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return PC.GpuBackend()


def ml_case(dataset, mnph, n, seed=3):
    """``n`` training links of the MovieLens-shaped graph bench.py runs on, as a parity case without reference
    records (the reference's own extractor needs seconds per link at this size; sizes / candidate sets / induced
    edges are checked by ``parity_checks.check_sampled``)."""
    from igmc_amd import preprocessing
    split = preprocessing.create_trainvaltest_split(dataset, 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
    pick = np.random.default_rng(seed).permutation(len(tr_u))[:n]
    links = np.stack([tr_u[pick], tr_v[pick]], 1).astype(np.int64)
    return dict(A=A, links=links, link_labels=np.asarray(tr_l)[pick].astype(np.int64),
                class_values=np.asarray(cv, dtype=np.float64), h=1, sample_ratio=1.0, mnph=mnph, recs=[None] * n)


@pytest.fixture(scope='module')
def ml1m():
    return ml_case('ml_1m', 100, 250)


def monti_case(name, n, seed=2):
    """``n`` training links of a bundled real dataset (uncapped: the reference default --max-nodes-per-hop 10000)."""
    from igmc_amd import preprocessing
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti(name, testing=True)
    pick = np.random.default_rng(seed).permutation(len(tr_u))[:n]
    return dict(A=A, links=np.stack([tr_u[pick], tr_v[pick]], 1).astype(np.int64),
                link_labels=np.asarray(tr_l)[pick].astype(np.int64), class_values=np.asarray(cv, dtype=np.float64),
                h=1, sample_ratio=1.0, mnph=10000, recs=[None] * n)


def first(case, n, start=0):
    c = dict(case)
    c['links'], c['link_labels'], c['recs'] = case['links'][start:start + n], case['link_labels'][start:start + n], [None] * n
    return c


@pytest.mark.parametrize('drop', [False, True])
def test_headline_batch_matches_oracle(be, ml1m, drop, monkeypatch, capfd):
    """ml_1m shape, batch 50, cap 100: the production launch (cluster of 4 workgroups per subgraph) vs the oracle."""
    monkeypatch.setenv('IGMC_GRAPH_STEP', '1')
    monkeypatch.setenv('IGMC_GS_TRACE', '1')
    case = first(ml1m, 50)
    res = PC.run_model_parity(be, case, R=5, use_dropout=drop)
    assert res['worst_grad_err'] < PC.GRAD_TOL
    d = res['d']
    assert d['B'] == 50 and 50 * 150 < d['N'] <= 50 * 202 and d['E'] > 150000       # the headline shape, not a toy
    assert res['ws'].step_form(res['batch'], 50) == 1         # (igmc_model_step_form: what the pipeline paces its extraction by)
    PC.check_sampled(d, case)
    err = capfd.readouterr().err
    assert 'k_graph_step B=50 train=1' in err and 'cluster=4' in err, err[-400:]


def test_headline_fused_train_steps_track_torch_adam(be, ml1m, monkeypatch, capfd):
    """5 steps of igmc_train_step on 5 different batches of 50 (k_tail_ts + k_finalize_ts write the weights)."""
    monkeypatch.setenv('IGMC_GRAPH_STEP', '1')
    monkeypatch.setenv('IGMC_GS_TRACE', '1')
    res = PC.run_fused_train_trajectory(be, ml1m, R=5, steps=5, batch=50)
    err = capfd.readouterr().err
    assert err.count('k_graph_step B=50 train=1') >= 5 and 'cluster=4' in err
    assert res['frac_off'] < PC.TRAJ_FRAC_OFF
    # sum over steps of loss * num_graphs (reference train_eval.py:176)
    assert res['total'] == pytest.approx(sum(50 * l for l, _ in res['losses']), rel=1e-5)


@pytest.mark.parametrize('name,n,R,mnph,drop', [
    ('ml1m', 250, 5, 100, True),            # subgraph kernel with edge flags
    ('douban', 250, 5, None, True),         # per-layer kernels + k_finalize (uncapped)
    ('flixster', 250, 10, None, False),     # R = 10: relation groups on the dense layers
    ('yahoo_music', 100, 71, None, True),   # R = 71: row walkers + k_finalize
])
def test_fused_train_steps_other_paths(be, ml1m, name, n, R, mnph, drop):
    case = ml1m if name == 'ml1m' else monti_case(name, n)
    res = PC.run_fused_train_trajectory(be, case, R=R, steps=5, batch=n // 5, use_dropout=drop)
    assert res['frac_off'] < PC.TRAJ_FRAC_OFF


@pytest.mark.parametrize('lean', [False, True])
@pytest.mark.parametrize('force_undirected', [False, True])
@pytest.mark.parametrize('name', ['ml1m', 'douban'])
def test_free_running_dropout(be, ml1m, name, force_undirected, lean):
    # lean: the arena keeps the dense blocks only and the draws are taken on them (k_relm_dropout); the CSR read back by
    # the check takes its flags from the blocks.  douban is uncapped: its slots are bounded by the longest row / column.
    case = first(ml1m, 50, 100) if name == 'ml1m' else monti_case('douban', 40)
    res = PC.run_free_running_dropout(be, case, R=5, p=0.2, force_undirected=force_undirected, lean=lean)
    assert res['worst_grad_err'] < PC.GRAD_TOL


def test_uncapped_douban_takes_the_subgraph_kernel(be):
    """The reference's default for the Monti datasets is no per-hop cap (--max-nodes-per-hop 10000): douban's longest row /
    column (< 128) bounds the slots, so the matrix-core subgraph kernel takes its batches of 50."""
    from igmc_amd import engine
    case = monti_case('douban', 50)
    g, b, d = PC.extract_case(be, case, replay=False)
    ws = engine.ModelWorkspace(be.lib, be.device, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    assert ws.dense_path(b, 50)
    res = PC.run_model_parity(be, case, R=5, use_dropout=True)
    assert res['worst_grad_err'] < PC.GRAD_TOL




@pytest.mark.parametrize('drop,lean', [(False, False), (True, True)])
def test_flixster_ten_relations_take_the_dense_layers(be, drop, lean):
    """flixster (ten rating levels, reference Main.py:387; slots of up to 155 items a side): k_dl_fwd / k_dl_bwd take the
    relations in two groups of five on the matrix cores; forward, loss and every gradient vs the oracle."""
    case = monti_case('flixster', 50)
    res = PC.run_model_parity(be, case, R=10, use_dropout=drop, lean=lean)
    assert res['worst_grad_err'] < PC.GRAD_TOL
    assert res['batch'].dense_layers(res['ws'])


@pytest.mark.parametrize('force_undirected', [False, True])
def test_flixster_free_running_dropout_on_the_dense_blocks(be, force_undirected):
    case = monti_case('flixster', 40)
    res = PC.run_free_running_dropout(be, case, R=10, p=0.2, force_undirected=force_undirected, lean=True)
    assert res['worst_grad_err'] < PC.GRAD_TOL


def test_flixster_fused_train_steps_are_bit_reproducible(be):
    """Five fused steps on the relation-group kernels, twice from the same state: the same bits (fixed-order reductions)."""
    case = monti_case('flixster', 250)
    runs = [PC.run_fused_train_trajectory(be, case, R=10, steps=5, batch=50, use_dropout=True) for _ in range(2)]
    assert runs[0]['frac_off'] < PC.TRAJ_FRAC_OFF
    for k in ('params', 'm1', 'm2'):
        assert np.array_equal(runs[0][k], runs[1][k]), k


def ml10m_case(n, seed=4):
    """The R = 10 counterpart of the headline shape (bench.py's ``ml_10m_lite``: ML-10M's ten half-star levels on the
    ml_1m-shaped graph, cap 100; reference Main.py:155-163 lists ml_10m beside flixster)."""
    from igmc_amd import preprocessing
    rmap = {float(i): i / 2.0 for i in range(1, 11)}
    split = preprocessing.create_trainvaltest_split('ml_10m_lite', 1234, True, rating_map=rmap, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
    pick = np.random.default_rng(seed).permutation(len(tr_u))[:n]
    links = np.stack([tr_u[pick], tr_v[pick]], 1).astype(np.int64)
    return dict(A=A, links=links, link_labels=np.asarray(tr_l)[pick].astype(np.int64),
                class_values=np.asarray(cv, dtype=np.float64), h=1, sample_ratio=1.0, mnph=100, recs=[None] * n)


@pytest.fixture(scope='module')
def ml10m():
    return ml10m_case(250)


@pytest.mark.parametrize('drop,lean', [(False, True), (True, False), (True, True)])
def test_ml10m_lite_batch_matches_oracle(be, ml10m, drop, lean):
    """Ten relations x cap 100 x batch 50 (E ~ 270 k): two workgroups per side AND two relation groups on k_dl_fwd /
    k_dl_bwd -- the product of what flixster (relation groups at 49 x 155 slots) and the cap-100 cases (R = 5) cover
    separately, and a shape bench.py times (``secondary.ml_10m_lite``).  Forward, loss, every gradient vs the oracle."""
    case = first(ml10m, 50)
    assert len(case['class_values']) == 10
    res = PC.run_model_parity(be, case, R=10, use_dropout=drop, lean=lean)
    assert res['worst_grad_err'] < PC.GRAD_TOL
    assert res['batch'].dense_layers(res['ws'])
    assert res['ws'].step_form(res['batch'], 50) == 3         # dense-layer kernels, both relation groups at once
    d = res['d']
    assert d['B'] == 50 and 50 * 150 < d['N'] <= 50 * 202 and d['E'] > 150000
    assert int(d['erel'].max()) >= 8                   # relation codes of the second group are present
    PC.check_sampled(d, case)


def test_ml10m_lite_free_running_dropout_on_the_dense_blocks(be, ml10m):
    res = PC.run_free_running_dropout(be, first(ml10m, 50, 50), R=10, p=0.2, force_undirected=False, lean=True)
    assert res['worst_grad_err'] < PC.GRAD_TOL


def test_ml10m_lite_fused_train_steps_track_torch_adam_and_are_bit_reproducible(be, ml10m):
    """Five fused steps (k_dl_fwd -> k_dl_bwd -> k_tail_ts -> k_finalize_ts with the group images of the next step) on five
    batches of 50 vs pyg_ref.train_step + torch Adam, twice from the same state: the same bits."""
    runs = [PC.run_fused_train_trajectory(be, ml10m, R=10, steps=5, batch=50, use_dropout=True) for _ in range(2)]
    assert runs[0]['frac_off'] < PC.TRAJ_FRAC_OFF
    for k in ('params', 'm1', 'm2'):
        assert np.array_equal(runs[0][k], runs[1][k]), k


@pytest.mark.parametrize('lean', [False, True])
def test_ml100k_cap200_batch_matches_oracle(be, lean):
    """BASELINE.json configs[1]: ml_100k shape, cap 200, adj-dropout 0.2, batch 50 -- slots of 201 nodes a side: the dense
    per-layer kernels (k_dl_layer0 / k_dl_layer on the matrix cores; lean: no edge list anywhere in the step)."""
    case = ml_case('ml_100k', 200, 50, seed=5)
    res = PC.run_model_parity(be, case, R=5, use_dropout=True, lean=lean)
    assert res['worst_grad_err'] < PC.GRAD_TOL
    assert res['batch'].dense_layers(res['ws'])
    assert res['ws'].step_form(res['batch'], 50) == 2         # dense-layer kernels, one relation group
    PC.check_sampled(res['d'], case)
    assert res['d']['N'] > 50 * 200


@pytest.mark.parametrize('force_undirected', [False, True])
def test_ml100k_cap200_free_running_dropout_on_the_dense_blocks(be, force_undirected):
    """Edge dropout drawn on the dense blocks of a lean cap-200 arena (k_relm_dropout keeps the transposed copy in step):
    flags vs the host restatement of the hash, model vs the oracle with those flags."""
    case = ml_case('ml_100k', 200, 50, seed=6)
    res = PC.run_free_running_dropout(be, case, R=5, p=0.2, force_undirected=force_undirected, lean=True)
    assert res['worst_grad_err'] < PC.GRAD_TOL


def test_ml100k_cap200_fused_train_steps_track_torch_adam(be):
    """... and five fused train steps (igmc_train_step on the dense per-layer path) vs pyg_ref.train_step + torch Adam."""
    case = ml_case('ml_100k', 200, 50, seed=7)
    res = PC.run_fused_train_trajectory(be, case, R=5, steps=5, batch=10, use_dropout=True)
    assert res['frac_off'] < PC.TRAJ_FRAC_OFF


def test_ml100k_cap200_fused_train_steps_are_bit_reproducible(be):
    """The same fused steps, run several times from the same state, give the same bits: every reduction on the path has
    a fixed order (no float atomics).  Guards the d att partials of k_dl_layer: compiled as packed-f32 chains they were
    NOT reproducible on gfx950 (profiles/r02_dl_packed_f32_nondeterminism.txt)."""
    case = ml_case('ml_100k', 200, 50, seed=7)
    first = None
    for i in range(6):
        res = PC.run_fused_train_trajectory(be, case, R=5, steps=2, batch=10, use_dropout=True)
        if first is None:
            first = res
            continue
        for k in ('params', 'm1', 'm2'):
            assert np.array_equal(first[k], res[k]), (i, k, float(np.abs(first[k] - res[k]).max()))


def test_ml100k_cap200_launch_forms_agree_bit_for_bit(be, monkeypatch):
    """The dense layers of the config-2 step as ONE launch per direction (k_dl_fwd / k_dl_bwd: the members of a subgraph
    exchange h_l / dPre_l through tagged words), the forward only as one launch, or one launch per layer pass: the same
    arithmetic in the same order -- parameters and Adam moments after the steps are bit-identical.  (The G / Y form of the
    backward, IGMC_DL_TS=0 -- what the sort-pool family and igmc_model_backward run -- sums in another order: within the
    trajectory tolerances of the oracle.)"""
    case = ml_case('ml_100k', 200, 50, seed=7)
    runs = {}
    for fused in ('2', '1', '0'):
        monkeypatch.setenv('IGMC_DL_FUSED', fused)
        runs[fused] = PC.run_fused_train_trajectory(be, case, R=5, steps=3, batch=10, use_dropout=True)
    monkeypatch.delenv('IGMC_DL_FUSED')
    runs['default'] = PC.run_fused_train_trajectory(be, case, R=5, steps=3, batch=10, use_dropout=True)
    for other in ('1', '0', 'default'):
        for k in ('params', 'm1', 'm2'):
            assert np.array_equal(runs['2'][k], runs[other][k]), (other, k, float(np.abs(runs['2'][k] - runs[other][k]).max()))
    for form, env in (('G / Y backward', 'IGMC_DL_TS'),):
        monkeypatch.setenv(env, '0')
        res = PC.run_fused_train_trajectory(be, case, R=5, steps=3, batch=10, use_dropout=True)
        monkeypatch.delenv(env)
        assert res['frac_off'] < PC.TRAJ_FRAC_OFF, form
        assert float(np.abs(res['params'] - runs['2']['params']).max()) < PC.TRAJ_P_MAX, form


def test_training_steps_survive_a_competing_full_chip_kernel(ml1m):
    """The cluster exchange of the subgraph kernel needs its members resident together.  Members are consecutive
    workgroups, so a chip that is partly held by ANOTHER kernel (here: a stream of large GEMMs on a second stream, all 256
    CUs busy) only delays clusters -- complete ones finish and free their CUs -- and never strands them: the steps must
    produce bit-identical parameters to an undisturbed run and no bounded wait may time out (igmc_model_check)."""
    import torch
    from igmc_amd.models import IGMC
    from igmc_amd.stepgraph import StepGraph
    from igmc_amd.train_eval import FlatAdam
    from igmc_amd.util_functions import MyDynamicDataset
    A, links, labels, cv = ml1m['A'], ml1m['links'], ml1m['link_labels'], ml1m['class_values']
    ds = MyDynamicDataset('data/t/hog', A, (links[:, 0], links[:, 1]), labels, 1, 1.0, 100, None, None, cv, device=0, seed=1)
    perm = torch.arange(len(ds))
    out = {}
    for mode in ('quiet', 'loaded'):
        torch.manual_seed(3)
        model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.0,
                     seed=1).to('cuda')
        model.reset_parameters()
        opt = FlatAdam(model, lr=1e-3)
        sg = StepGraph(model, opt, ds, 50, 0.001, use_graph=False, overlap=True)
        assert sg.ws.dense_path(sg.arenas[0], 50)
        hog = torch.cuda.Stream()
        stop = None
        if mode == 'loaded':
            x = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
            with torch.cuda.stream(hog):
                for _ in range(40):              # ~ tens of milliseconds of full-chip GEMM queued beside the steps
                    x = (x @ x).clamp_(-1, 1)
            stop = x
        sg.begin_epoch(perm, 1)
        for _ in range(5):
            sg.step()
        sg.check()                               # raises if a bounded device-side wait timed out
        torch.cuda.synchronize()
        out[mode] = model.flat_parameters().detach().cpu().clone()
        del stop
    assert torch.equal(out['quiet'], out['loaded'])


def _trajectory(ds, drop, perm, epochs=2, **sg_kw):
    """Two epochs of StepGraph on ``ds``: parameters, Adam moments, epoch totals (bit-comparable)."""
    import torch
    from igmc_amd.models import IGMC
    from igmc_amd.stepgraph import StepGraph
    from igmc_amd.train_eval import FlatAdam
    torch.manual_seed(3)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=drop,
                 seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, **sg_kw)
    totals = []
    for ep in range(1, epochs + 1):
        t, n = sg.run_epoch(perm, ep)          # (run_epoch ends in check(): no stamp mismatch, no timed-out wait)
        totals.append(float(t.item()))
    torch.cuda.synchronize()
    return sg, (model.flat_parameters().detach().cpu().clone(), opt.exp_avg.detach().cpu().clone(),
                opt.exp_avg_sq.detach().cpu().clone(), totals, opt.t)


def _assert_same(a, b, what):
    import torch
    for i, (x, y) in enumerate(zip(a, b)):
        if torch.is_tensor(x):
            assert torch.equal(x, y), '%s: tensor %d differs in %d of %d elements, max |d| %.3e' % (
                what, i, int((x != y).sum()), x.numel(), float((x - y).abs().max()))
        else:
            assert x == y, '%s: %r != %r' % (what, x, y)


@pytest.mark.parametrize('data', ['ml_1m', 'douban'])
@pytest.mark.parametrize('drop', [0.0, 0.2])
def test_step_graph_is_bit_reproducible_and_structure_independent(ml1m, monkeypatch, drop, data):
    """The DEFAULT training structure (groups of steps per hipGraph launch, next group's extraction on a second stream,
    lean arenas, edge dropout drawn on the dense blocks inside the graph) at the headline shape and on douban:
    * run twice from the same seed -> parameters, Adam moments and epoch totals bit-equal (no float atomics, no race);
    * groups of 8 per launch == groups of 4 == every step launched eagerly on one stream: the same kernels on the same
      batches, whatever launches them.
    Two epochs of 24 steps each (1200 links): the first epoch of a process holds an eager first step + a re-grouping, both
    epochs hold whole groups replayed from the graph; what is left behind the pairs is an eagerly launched remainder (first
    epoch) or one more group as a single-group launch (second epoch, round 6)."""
    import torch
    from igmc_amd.util_functions import MyDynamicDataset
    if data == 'ml_1m':
        A, cv = ml1m['A'], ml1m['class_values']
        rng = np.random.default_rng(11)
        coo = A.tocoo()
        pick = rng.permutation(coo.nnz)[:1200]
        u, v, y = coo.row[pick], coo.col[pick], (coo.data[pick] - 1).astype(np.int64)
        ds = MyDynamicDataset('data/t/repro', A, (u, v), y, 1, 1.0, 100, None, None, cv, device=0, seed=1)
    else:
        from igmc_amd import preprocessing
        (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('douban', testing=True)
        ds = MyDynamicDataset('data/t/repro_d', A, (tr_u[:1200], tr_v[:1200]), tr_l[:1200], 1, 1.0, 10000, None, None, cv,
                              device=0, seed=1)
    perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
    sg, ref = _trajectory(ds, drop, perm, group=8)
    assert sg.ws.dense_path(sg.arenas[0], 50) and any(g is not None for g in sg.graphs) and ref[4] == 48
    # (the second epoch's 24 steps = a pair of groups of 8 + ONE single-group launch of the third group; the first epoch's
    #  23 steps behind its eager first one = a pair + 7 eager steps)
    assert sg.graph is not None and sg.graph1[0] is not None and sg.graph1[1] is None
    for rep in range(2):
        _, again = _trajectory(ds, drop, perm, group=8)
        _assert_same(ref, again, 'groups of 8, repeat %d' % rep)
    _, g4 = _trajectory(ds, drop, perm, group=4)
    _assert_same(ref, g4, 'groups of 4 vs groups of 8')
    sg_e, eager = _trajectory(ds, drop, perm, use_graph=False, overlap=False, group=8)
    assert not any(g is not None for g in sg_e.graphs)
    _assert_same(ref, eager, 'eager one-stream launches vs groups of 8')
    # ... however the next group's extraction launches are held back beside the steps: gate kernels polling the step counter
    # (the default here), edges out of the step chain, or not at all
    for mode in ('1', '0'):
        monkeypatch.setenv('IGMC_EXTRACT_PACED', mode)
        _, pm = _trajectory(ds, drop, perm, group=8)
        _assert_same(ref, pm, 'extraction pacing mode %s vs the gates' % mode)
    monkeypatch.delenv('IGMC_EXTRACT_PACED')
    # gates that KEEP giving up (here: a timeout of zero -- what a profiler that serialises the dispatches does to them at 2 ms
    # apiece) are counted in the control block; the epoch's check() reads the count and the object paces by edges from then on
    from igmc_amd.stepgraph import StepGraph
    monkeypatch.setattr(StepGraph, 'GATE_TIMEOUT_US', 0.0)
    sg_t, pt = _trajectory(ds, drop, perm, group=8)
    assert sg_t.pacing_fallback in (None, '1')        # (how many gates had to wait at all depends on the box's timing)
    _assert_same(ref, pt, 'gates timing out, then edges, vs the gates')
    from igmc_amd import _lib
    sg_t.pacing_fallback = None
    sg_t.ctrl[_lib.CTRL['GATE_TIMEOUTS']] = 7
    sg_t.check()
    assert sg_t.pacing_fallback == '1' and sg_t.graph is None
    assert int(sg_t.ctrl[_lib.CTRL['GATE_TIMEOUTS']].item()) == 0
    monkeypatch.setattr(StepGraph, 'GATE_TIMEOUT_US', 2000.0)
    # ... == the data-parallel step (igmc_train_step_dp) on a one-rank RCCL communicator: the subgraph kernel's tables and
    # the lin gradients go through a grouped all-reduce captured between k_tail_ts and k_finalize_ts -- a sum over one rank
    monkeypatch.setenv('IGMC_FORCE_DP_PATH', '1')
    monkeypatch.setenv('IGMC_DP_ALLREDUCE_ALWAYS', '1')
    sg_d, dp = _trajectory(ds, drop, perm, group=8)
    assert sg_d.dp_path and sg_d.comm is not None and sg_d.comm.info() == (0, 1)
    _assert_same(ref, dp, 'data-parallel step on a one-rank communicator vs the single-GPU step')


def test_step_graph_on_the_dense_per_layer_path_is_reproducible():
    """... and where the dense per-layer kernels take the step (config 2 shape: ml_100k, cap 200, edge dropout 0.2)."""
    import torch
    from igmc_amd import preprocessing
    from igmc_amd.util_functions import MyDynamicDataset
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.create_trainvaltest_split('ml_100k', 1234, True, verbose=False)
    pick = np.random.default_rng(3).permutation(len(tr_u))[:1000]
    ds = MyDynamicDataset('data/t/frdl', A, (tr_u[pick], tr_v[pick]), np.asarray(tr_l)[pick], 1, 1.0, 200, None, None, cv,
                          device=0, seed=1)
    perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
    sg, ref = _trajectory(ds, 0.2, perm, group=8)
    assert sg.arenas[0].dense_layers(sg.ws) and any(g is not None for g in sg.graphs)
    _, again = _trajectory(ds, 0.2, perm, group=8)
    _assert_same(ref, again, 'dense per-layer path, repeat')
    _, eager = _trajectory(ds, 0.2, perm, use_graph=False, overlap=False, group=8)
    _assert_same(ref, eager, 'dense per-layer path, eager vs graph')


# ---------------------------------------------------------------------- against the reference's OWN models.py / train_eval.py
@pytest.mark.parametrize('tag', ['nodrop', 'drop'])
def test_headline_batch_matches_the_reference_fixture(be, tag, monkeypatch, capfd):
    """BASELINE.json configs[2] shape (ml_1m-shaped graph, cap 100, ONE batch of 50) against the fixture the UNMODIFIED
    reference ``models.py`` / ``train_eval.py`` produced (``tests/golden/make_model_golden.py``): the node lists its
    extractor chose are replayed, then eval outputs, train outputs, the step's loss (= the epoch loss its ``train``
    returned), EVERY gradient, and the parameters after its ``train`` + Adam step -- tolerances of this file's oracle
    checks (outputs 2e-5, loss 3e-6, gradients 5e-5 of the tensor's peak)."""
    from helpers import load_model_golden
    monkeypatch.setenv('IGMC_GRAPH_STEP', '1')
    monkeypatch.setenv('IGMC_GS_TRACE', '1')
    res = PC.run_reference_fixture(be, load_model_golden('headline_' + tag), 50)
    assert res['N'] > 50 * 150 and res['E'] > 150000
    err = capfd.readouterr().err
    assert 'k_graph_step B=50 train=1' in err and 'cluster=4' in err, err[-400:]


@pytest.mark.parametrize('case', ['igmc_r5', 'igmc_side', 'igmc_r10'])
def test_small_batches_match_the_reference_fixtures(be, case):
    """R = 5 with edge dropout, side features + multiply_by + force_undirected, R = 10 (flixster): three batches of 8
    each, the whole epoch of the reference's ``train`` (loss bookkeeping + Adam) through the fused step."""
    from helpers import load_model_golden
    res = PC.run_reference_fixture(be, load_model_golden(case), 8)
    assert res['params_frac_off'] < PC.TRAJ_FRAC_OFF


def test_extraction_bitmaps_of_a_200k_node_graph(be, monkeypatch):
    """A rating graph of 170 000 users x 40 000 items: the LDS bitmaps of an extraction workgroup take 105 KB (the limit is
    ~300 k ids, igmc_batch_create) -- far from the 20 KB of the MovieLens shapes the other tests run on.  The single-batch
    launch against the host twin on hub links where the cap binds; then one epoch of the grouped step graph
    (k_extract_nodes_set: what the training pipeline uses) == the same epoch with one batch per extraction launch, bit for bit."""
    import scipy.sparse as ssp
    import torch
    from igmc_amd.util_functions import MyDynamicDataset
    from oracle import extract_cpu
    import test_extract_twin as T
    rng = np.random.default_rng(23)
    nu, nv = 170000, 40000
    u = rng.integers(0, nu, 500000)
    v = rng.integers(0, nv, 500000)
    hub_users = rng.choice(nu, 3000, replace=False)          # item 7: rated by 3 000 users
    hub_items = rng.choice(nv, 2500, replace=False)          # user 11: rates 2 500 items
    u = np.concatenate([u, hub_users, np.full(2500, 11)])
    v = np.concatenate([v, np.full(3000, 7), hub_items])
    r = rng.integers(1, 6, len(u)).astype(np.float32)
    A = ssp.csr_matrix(ssp.coo_matrix((r, (u, v)), shape=(nu, nv)))
    A.data = np.clip(np.rint(A.data), 1, 5).astype(np.float32)           # (duplicate pairs were summed)
    cv = np.arange(1, 6, dtype=np.float64)
    coo = A.tocoo()
    hubs = np.flatnonzero((coo.col == 7) | (coo.row == 11))
    pick = np.concatenate([rng.choice(hubs, 150, replace=False), rng.permutation(coo.nnz)[:250]])
    lu, lv, ly = coo.row[pick].astype(np.int64), coo.col[pick].astype(np.int64), (coo.data[pick] - 1).astype(np.int64)
    # single-batch launch (k_extract_nodes) against the OpenMP twin: node sets, order, labels, induced edges
    case = dict(A=A, links=np.stack([lu[:24], lv[:24]], 1), link_labels=ly[:24], class_values=cv, h=1, sample_ratio=1.0,
                mnph=100, recs=[None] * 24)
    _, _, d = PC.extract_case(be, case, replay=False, seed=11, epoch=4)
    twin = extract_cpu.extract_batch(A, case['links'][:, 0], case['links'][:, 1], 0, 24, hop=1, sample_ratio=1.0,
                                     max_nodes_per_hop=100, seed=11, epoch=4)
    T.compare(d, case, twin)
    assert max(len(t[0]) for t in twin) == 101 and max(len(t[1]) for t in twin) == 101          # the cap binds
    # the training pipeline: group launches (k_extract_nodes_set) == one batch per launch
    ds = MyDynamicDataset('data/t/big', A, (lu, lv), ly, 1, 1.0, 100, None, None, cv, device=0, seed=1)
    perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
    sg, ref = _trajectory(ds, 0.0, perm, epochs=1, group=4)
    assert sg.ws.dense_path(sg.arenas[0], 50) and ref[4] == 8
    monkeypatch.setenv('IGMC_GROUP_EXTRACT_CHUNK', '1')
    _, single = _trajectory(ds, 0.0, perm, epochs=1, group=4)
    _assert_same(ref, single, 'one batch per extraction launch vs group launches on a 210 000-node graph')


def test_prefetched_arena_after_a_ragged_batch_keeps_its_size():
    """An epoch of 29 full batches + 6 links in groups of 8: a pair of groups, ONE single-group launch (which leaves the odd
    arena set current), eager steps on that set, the ragged batch in one of its arenas.  In the next epoch the same arena is
    filled inside a REPLAYED launch and consumed by an eager step -- which took its batch size from the arena's last extraction
    CALL on the host, the ragged one (found in round 6: epochs of douban / flixster with groups of 50 drifted from the eager
    order from the second epoch on).  Three epochs, replayed launches == eager launches, bit for bit."""
    import torch
    from igmc_amd import preprocessing
    from igmc_amd.util_functions import MyDynamicDataset
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.load_data_monti('douban', testing=True)
    n = 29 * 50 + 6
    ds = MyDynamicDataset('data/t/ragged_d', A, (tr_u[:n], tr_v[:n]), tr_l[:n], 1, 1.0, 10000, None, None, cv, device=0, seed=1)
    perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
    sg, ref = _trajectory(ds, 0.2, perm, epochs=3, group=8)
    assert sg.graph is not None and sg.graph1[0] is not None and ref[4] == 3 * 30
    _, eager = _trajectory(ds, 0.2, perm, epochs=3, use_graph=False, overlap=False, group=8)
    _assert_same(ref, eager, 'replayed launches vs eager launches over three epochs with a ragged last batch')
