"""``oracle/model_cpu.c`` -- an independent OpenMP host implementation of the IGMC model step (the ``igmc_cpu_*``
twins; subgraph-per-thread, aggregate-then-transform) -- against the oracle (``oracle/pyg_ref.py``: per-edge PyG formulation,
pinned to the reference's own model code) and against the HIP path (emulator here, MI355X under ``-m gpu``): outputs, loss,
every gradient tensor, Adam, and a five-step trajectory fed by the extraction twin.  Tolerances: those of the engine-vs-oracle
parity (``parity_checks.OUT_TOL`` / ``LOSS_RTOL`` / ``GRAD_TOL``)."""
import numpy as np
import pytest
import torch

import parity_checks as PC
from helpers import batch_to_pyg, load_extract_golden
from oracle import extract_cpu, model_cpu, pyg_ref

CASES = load_extract_golden()
REL = dict(hand=5, synth_cap=5, synth_nocap=5, douban=5, douban_cap20=5, flixster=10, flixster_h2=10, hand_h2=5,
           synth_h2_ratio=5, yahoo_music=71)


def worst_grad(got, ref):
    return max(float(np.abs(got[k] - v).max() / max(np.abs(v).max(), 1e-6)) for k, v in ref.items())


def pyg_of_raw(cb, L, y):
    """The twin's collated arrays as the oracle's batch."""
    class Bt(object):
        pass
    b = Bt()
    N = int(cb['node_off'][-1])
    x = np.zeros((N, L), np.float32)
    x[np.arange(N), cb['label']] = 1.0
    b.x = torch.from_numpy(x)
    b.edge_index = torch.from_numpy(np.stack([cb['src'], cb['dst']]).astype(np.int64))
    b.edge_type = torch.from_numpy(cb['rel'].astype(np.int64))
    b.batch = torch.from_numpy(np.repeat(np.arange(cb['B']), np.diff(cb['node_off'])).astype(np.int64))
    b.y = torch.from_numpy(np.asarray(y, np.float32))
    b.num_graphs = cb['B']
    return b


def twin_batch(case, first, B, seed=1, epoch=1):
    A = extract_cpu.prepare(case['A'])
    raw = extract_cpu.extract_batch(A, case['links'][:, 0], case['links'][:, 1], first, B, hop=case['h'],
                                    sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'], seed=seed, epoch=epoch, raw=True)
    return model_cpu.collate_raw(raw, raw[2].shape[1], raw[3].shape[1])


@pytest.mark.parametrize('name', sorted(REL))
@pytest.mark.parametrize('drop', [False, True])
def test_twin_matches_the_oracle(name, drop):
    """Batches made by the extraction twin; identical weights and dropout masks on both sides."""
    torch.set_num_threads(1)
    case, R = CASES[name], REL[name]
    L = 2 * case['h'] + 2
    B = min(6, len(case['links']))
    cb = twin_batch(case, 0, B)
    y = np.asarray(case['class_values'], np.float64)[np.asarray(case['link_labels'][:B])].astype(np.float32)
    pyg = pyg_of_raw(cb, L, y)
    mult = 1.0 if name != 'douban' else 2.5
    ref = PC.make_ref_model(L, R, seed=3, adj_dropout=0.2 if drop else 0.0, multiply_by=mult, fast=False)
    cfg = model_cpu.config_of(ref)
    flat = model_cpu.flat_from_model(ref, cfg)
    rng = np.random.default_rng(7)
    lin_mask = rng.random((B, 128)) < 0.5
    E = int(cb['edge_off'][-1])
    keep = (rng.random(E) >= 0.2) if drop else None
    ARR = 0.0 if name == 'synth_nocap' else 0.001
    rl, ro, rg = pyg_ref.loss_and_grads(ref, pyg, ARR=ARR, edge_mask=None if keep is None else torch.from_numpy(keep),
                                        lin_mask=torch.from_numpy(lin_mask))
    cbk = cb if keep is None else model_cpu.collate_pyg(pyg, keep)
    out, grad, loss = model_cpu.loss_grad(cfg, flat, cbk, y=y, lin_mask=lin_mask, multiply_by=mult, ARR=ARR)
    assert PC.rel_err(out, ro.numpy()) < PC.OUT_TOL
    assert loss[0] == PC.pytest_approx(float(rl), PC.LOSS_RTOL)
    assert worst_grad(model_cpu.unflatten(cfg, grad), {k: v.numpy() for k, v in rg.items()}) < PC.GRAD_TOL
    # eval mode: no dropout of either kind (reference train_eval.py:182-199)
    sse, eo = pyg_ref.eval_sse(ref, pyg)
    out2, g2, l2 = model_cpu.loss_grad(cfg, flat, cb, y=y, multiply_by=mult, want_grad=False)
    assert g2 is None and PC.rel_err(out2, eo.numpy()) < PC.OUT_TOL and l2[1] == PC.pytest_approx(sse, 1e-5)


def test_twin_is_deterministic_per_thread_count_and_close_across():
    case = CASES['douban']
    cb = twin_batch(case, 0, 8)
    y = np.linspace(1, 5, 8).astype(np.float32)
    ref = PC.make_ref_model(4, 5, seed=5)
    cfg = model_cpu.config_of(ref)
    flat = model_cpu.flat_from_model(ref, cfg)
    lm = np.random.default_rng(1).random((8, 128)) < 0.5
    runs = {}
    for th in (1, 3, 3, 1):
        assert model_cpu.set_threads(th) == th
        runs.setdefault(th, []).append(model_cpu.loss_grad(cfg, flat, cb, y=y, lin_mask=lm))
    model_cpu.set_threads(4)
    for th in (1, 3):
        (o0, g0, l0), (o1, g1, l1) = runs[th]
        assert np.array_equal(o0, o1) and np.array_equal(g0, g1) and l0 == l1
    assert np.array_equal(runs[1][0][0], runs[3][0][0])                   # (outputs do not depend on the reduction)
    assert np.abs(runs[1][0][1] - runs[3][0][1]).max() <= 1e-5 * np.abs(runs[1][0][1]).max()


def test_twin_adam_tracks_torch_adam():
    rng = np.random.default_rng(2)
    n = 5000
    p0 = rng.standard_normal(n).astype(np.float32)
    for wd in (0.0, 0.01):
        tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
        opt = torch.optim.Adam([tp], lr=1e-3, weight_decay=wd)
        p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        for t in range(1, 6):
            g = rng.standard_normal(n).astype(np.float32)
            tp.grad = torch.from_numpy(g.copy())
            opt.step()
            model_cpu.adam_step(p, g, m, v, t, lr=1e-3, weight_decay=wd)
            assert np.abs(p - tp.detach().numpy()).max() < 2e-7


def test_twin_train_steps_track_the_oracle():
    """Extraction twin -> model twin -> Adam twin, five batches, against pyg_ref.train_step + torch.optim.Adam."""
    torch.set_num_threads(1)
    case = CASES['douban']
    B, L, R = 6, 4, 5
    ref = PC.make_ref_model(L, R, seed=9, fast=False)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    cfg = model_cpu.config_of(ref)
    flat = model_cpu.flat_from_model(ref, cfg)
    m, v = np.zeros_like(flat), np.zeros_like(flat)
    rng = np.random.default_rng(4)
    nb = len(case['links']) // B
    assert nb >= 2
    for s in range(5):
        f = (s % nb) * B
        cb = twin_batch(case, f, B, epoch=1 + s // nb)
        y = np.asarray(case['class_values'], np.float64)[np.asarray(case['link_labels'][f:f + B])].astype(np.float32)
        lm = rng.random((B, 128)) < 0.5
        _, grad, loss = model_cpu.loss_grad(cfg, flat, cb, y=y, lin_mask=lm)
        model_cpu.adam_step(flat, grad, m, v, s + 1)
        rl = pyg_ref.train_step(ref, opt, pyg_of_raw(cb, L, y), ARR=0.001, lin_mask=torch.from_numpy(lm))
        assert loss[0] == PC.pytest_approx(rl, PC.TRAJ_LOSS_RTOL)
    # the same yardstick as the engine's trajectories (parity_checks.run_fused_train_trajectory): exp_avg is linear in the
    # gradients -> tight; an element whose gradient is float noise may step the other way -> a bounded fraction off
    want = model_cpu.flat_from_model(ref, cfg)
    named = dict(ref.named_parameters())
    for key, off, shape in cfg.layout():
        ea = opt.state[named[key]]['exp_avg'].numpy().reshape(-1)
        assert np.abs(m[off:off + ea.size] - ea).max() <= PC.TRAJ_M1_TOL * max(np.abs(ea).max(), 1e-9), key
    diff = np.abs(flat - want)
    assert (diff > PC.TRAJ_P_ATOL + PC.TRAJ_P_RTOL * np.abs(want)).mean() < PC.TRAJ_FRAC_OFF and diff.max() <= PC.TRAJ_P_MAX


@pytest.mark.parametrize('name', ['hand', 'synth_cap', 'flixster'])
def test_hip_path_on_the_emulator_matches_the_twin(name):
    """Engine forward / loss / gradients (kernels on the emulator) against the twin on the engine's own batch and masks."""
    be = PC.EmuBackend()
    res = PC.run_model_parity(be, CASES[name], R=REL[name], use_dropout=True, check_eval=False)
    compare_engine(res, REL[name])


def compare_engine(res, R):
    ref, pyg = res['ref'], res['pyg']
    cfg = model_cpu.config_of(ref)
    flat = model_cpu.flat_from_model(ref, cfg)
    cb = model_cpu.collate_pyg(pyg, res['edge_keep'])
    out, grad, loss = model_cpu.loss_grad(cfg, flat, cb, y=pyg.y.numpy(), lin_mask=res['lin_mask'])
    assert PC.rel_err(res['train_out'], out) < PC.OUT_TOL
    assert float(res['loss'][0]) == PC.pytest_approx(loss[0], PC.LOSS_RTOL)
    assert worst_grad(res['grads'], model_cpu.unflatten(cfg, grad)) < PC.GRAD_TOL
    # and the twin against the oracle on the same batch: three implementations, one answer
    rl, ro, rg = res['oracle']
    assert PC.rel_err(out, ro) < PC.OUT_TOL and worst_grad(model_cpu.unflatten(cfg, grad), rg) < PC.GRAD_TOL


@pytest.mark.gpu
@pytest.mark.parametrize('drop', [False, True])
def test_gpu_headline_batch_matches_the_twin(drop):
    import test_gpu_headline as H
    be = PC.GpuBackend()
    case = H.first(H.ml_case('ml_1m', 100, 50, seed=1), 50)
    res = PC.run_model_parity(be, case, R=5, use_dropout=drop, check_eval=False)
    assert res['d']['B'] == 50 and res['d']['E'] > 150000
    compare_engine(res, 5)


def test_twin_rejects_malformed_batches():
    case = CASES['hand']
    cb = twin_batch(case, 0, 2)
    ref = PC.make_ref_model(4, 5, seed=1)
    cfg = model_cpu.config_of(ref)
    flat = model_cpu.flat_from_model(ref, cfg)
    for key, val in (('label', 9), ('rel', 7), ('src', int(cb['node_off'][-1]))):
        bad = dict(cb)
        bad[key] = cb[key].copy()
        bad[key][0] = val
        with pytest.raises(RuntimeError):
            model_cpu.loss_grad(cfg, flat, bad, want_grad=False)
    no_target = dict(cb)
    no_target['label'] = np.maximum(cb['label'], 2).astype(np.int32)
    with pytest.raises(RuntimeError):
        model_cpu.loss_grad(cfg, flat, no_target, want_grad=False)
