"""Parity checks shared by the kernel-logic tests (host emulation of the HIP sources, CPU) and the
GPU tests (real gfx950 library through the same C ABI).  A ``Backend`` hides where buffers live."""
import ctypes

import numpy as np

from helpers import batch_to_pyg, golden_canonical, graph_canonical
from igmc_amd import engine


class EmuBackend(object):
    """numpy buffers + libigmc_emu.so (kernel logic on the CPU; test infrastructure only)."""
    name = 'emu'

    def __init__(self):
        from helpers import emu_lib
        self.lib = emu_lib()
        self.device = 0

    def dev(self, arr):
        return np.ascontiguousarray(arr)

    def ptr(self, buf):
        return buf.ctypes.data

    def host(self, buf):
        return np.array(buf, copy=True)

    def sync(self):
        pass


class GpuBackend(object):
    """torch CUDA(=HIP) tensors + libigmc_hip.so -- the product path."""
    name = 'gpu'

    def __init__(self):
        import torch
        from igmc_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        self.device = 0
        torch.cuda.set_device(0)

    def dev(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).cuda()

    def ptr(self, buf):
        return buf.data_ptr()

    def host(self, buf):
        self.torch.cuda.synchronize()
        return buf.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


def extract_case(be, case, replay, max_graphs=None, seed=0, epoch=0):
    """Run the engine extraction on a golden case; returns (graph, batch, downloaded dict)."""
    A = case['A']
    g = engine.Graph(A, device=be.device, lib=be.lib)
    B = len(case['recs'])
    b = engine.Batch(g, max_graphs=max_graphs or B, hop=case['h'], max_nodes_per_hop=case['mnph'])
    ys = case['class_values'][case['link_labels']].astype(np.float32)
    if replay:
        ul, vl, ud, vd = [], [], [], []
        for rec in case['recs']:
            nu = len(rec['u_nodes'])
            ul.append(rec['u_nodes'])
            vl.append(rec['v_nodes'])
            ud.append(rec['labels'][:nu] // 2)
            vd.append(rec['labels'][nu:] // 2)
        b.extract_replay(ul, vl, ud, vd, ys)
    else:
        lu = be.dev(case['links'][:, 0].astype(np.int32))
        lv = be.dev(case['links'][:, 1].astype(np.int32))
        ly = be.dev(ys)
        b.extract(be.ptr(lu), be.ptr(lv), be.ptr(ly), None, 0, B, sample_ratio=case['sample_ratio'], seed=seed,
                  epoch=epoch)
        be.sync()
    return g, b, b.download()


def check_batch_structure(d, num_labels):
    """Invariants of a collated batch, independent of any reference."""
    B, N, E = d['B'], d['N'], d['E']
    assert d['node_off'][0] == 0 and d['node_off'][B] == N
    assert d['row_ptr'][0] == 0 and d['row_ptr'][N] == E
    assert np.all(np.diff(d['row_ptr']) >= 0)
    assert np.all(d['node_label'] < num_labels)
    for g in range(B):
        lo, hi, nu = d['node_off'][g], d['node_off'][g + 1], d['n_users'][g]
        assert np.all(d['node_graph'][lo:hi] == g)
        assert d['node_label'][lo] == 0 and d['node_label'][lo + nu] == 1        # targets first
        assert np.all(d['node_label'][lo:lo + nu] % 2 == 0) and np.all(d['node_label'][lo + nu:hi] % 2 == 1)
        # non-target nodes in ascending id order (deterministic layout)
        assert np.all(np.diff(d['node_gid'][lo + 1:lo + nu]) > 0)
        assert np.all(np.diff(d['node_gid'][lo + nu + 1:hi]) > 0)
    # rows are sorted by relation (needed by the att-gradient run logic)
    for i in range(N):
        r = d['erel'][d['row_ptr'][i]:d['row_ptr'][i + 1]]
        assert np.all(np.diff(r.astype(np.int32)) >= 0)
    assert np.all(d['eflag'] == 3)


def check_against_golden(d, case):
    check_batch_structure(d, 2 * case['h'] + 2)
    assert d['B'] == len(case['recs'])
    for g, rec in enumerate(case['recs']):
        users, items, ulab, vlab, edges = graph_canonical(d, g)
        gun, gvn, gulab, gvlab, gt = golden_canonical(rec)
        assert users[0] == gun[0] and items[0] == gvn[0]
        assert sorted(users.tolist()) == sorted(gun.tolist())
        assert sorted(items.tolist()) == sorted(gvn.tolist())
        assert ulab == gulab and vlab == gvlab
        assert np.array_equal(edges, gt), 'graph %d: induced edges differ from the reference' % g
        assert d['y'][g] == np.float32(rec['y'])


def check_sampled(d, case):
    """Free-running sampler on a capped case: sizes match the reference's, nodes come from the right
    candidate sets, edges are exactly the induced edges of the chosen nodes (checked vs the oracle)."""
    from oracle import extract_ref as X
    A, Acsc = case['A'], case['A'].tocsc()
    check_batch_structure(d, 2 * case['h'] + 2)
    for g, rec in enumerate(case['recs']):
        users, items, ulab, vlab, edges = graph_canonical(d, g)
        assert len(users) == len(rec['u_nodes']) or case['h'] > 1
        assert len(items) == len(rec['v_nodes']) or case['h'] > 1
        i, j = case['links'][g]
        if case['h'] == 1:
            cand_u = set(Acsc.indices[Acsc.indptr[j]:Acsc.indptr[j + 1]].tolist()) - {i}
            cand_v = set(A.indices[A.indptr[i]:A.indptr[i + 1]].tolist()) - {j}
            assert set(users[1:].tolist()) <= cand_u and set(items[1:].tolist()) <= cand_v
        udist = [ulab[int(u)] // 2 for u in users]
        vdist = [vlab[int(v)] // 2 for v in items]
        out = X.subgraph_extraction_labeling((i, j), A, Acsc, case['h'], 1.0, None, None, None,
                                             case['class_values'], case['link_labels'][g],
                                             node_lists=(users, items, udist, vdist))
        ce = X.canonical_edges(users, items, out[0], out[1], out[2])
        assert np.array_equal(ce, edges)


# ====================================================================== model parity
def make_ref_model(L, R, n_side=0, seed=1, adj_dropout=0.0, multiply_by=1.0, fast=True):
    import torch
    from oracle import pyg_ref
    torch.manual_seed(seed)
    m = pyg_ref.IGMCRef(L, (32, 32, 32, 32), R, 4, adj_dropout=adj_dropout, side_features=n_side > 0,
                        n_side_features=n_side, multiply_by=multiply_by, fast=fast)
    # make biases / att non-trivial so that every gradient path is exercised
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m


def flatten_params(ws, model):
    flat = np.zeros(ws.n_params, np.float32)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for key, off, shape in ws.layout():
        a = sd[key].astype(np.float32).reshape(-1)
        assert a.size == int(np.prod(shape)), key
        flat[off:off + a.size] = a
    return flat


def unflatten_grads(ws, flat):
    return {key: flat[off:off + int(np.prod(shape))].reshape(shape) for key, off, shape in ws.layout()}


def reverse_positions(d):
    """rev[p] = CSR position of the reverse of the edge stored at p."""
    N = d['N']
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
    src = d['col'].astype(np.int64)
    pos = {(int(s), int(t)): p for p, (s, t) in enumerate(zip(src, dst))}
    return np.array([pos[(int(t), int(s))] for s, t in zip(src, dst)], dtype=np.int64)


def run_model_parity(be, case, R, ARR=0.001, use_dropout=True, multiply_by=1.0, seed=3, check_eval=True,
                     rtol=2e-4, atol=2e-5, n_side=0):
    """Engine forward / loss+grad on one extracted batch vs the PyG-1.4.2 restatement (oracle/pyg_ref.py)
    on IDENTICAL inputs: same subgraphs, same weights, same dropout masks."""
    import torch
    from oracle import pyg_ref
    g, b, d = extract_case(be, case, replay=False)
    L = 2 * case['h'] + 2
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, n_side, b.node_capacity, b.edge_capacity, b.max_graphs)
    ref = make_ref_model(L, R, n_side=n_side, seed=seed, adj_dropout=0.2 if use_dropout else 0.0,
                         multiply_by=multiply_by)
    flat = flatten_params(ws, ref)
    P = be.dev(flat)
    B = d['B']
    out = be.dev(np.zeros(B, np.float32))
    pyg = batch_to_pyg(d, L)
    side_buf = None
    if n_side:        # side features of the two target nodes (reference models.py:208-209)
        side = np.random.default_rng(seed + 1).standard_normal((B, n_side)).astype(np.float32)
        side_buf = be.dev(side)
        b.set_side_features(be.ptr(side_buf), n_side)
        pyg.u_feature = torch.from_numpy(side[:, :n_side // 2].copy())
        pyg.v_feature = torch.from_numpy(side[:, n_side // 2:].copy())
    res = {}
    if check_eval:
        ws.forward(be.ptr(P), b, be.ptr(out), training=False, multiply_by=multiply_by)
        sse, ref_out = pyg_ref.eval_sse(ref, pyg)
        got = be.host(out)
        np.testing.assert_allclose(got, ref_out.numpy(), rtol=rtol, atol=atol)
        res['eval_out'] = got
        acc = be.dev(np.zeros(2, np.float64))
        ws.sse_accumulate(be.ptr(out), b, be.ptr(acc))
        a = be.host(acc)
        assert a[1] == B and a[0] == pytest_approx(sse, 1e-4)
    # ---- training step with injected masks
    rng = np.random.default_rng(seed)
    lin_mask = (rng.random((B, 128)) < 0.5)
    edge_mask = None
    if use_dropout:
        keep = rng.random(d['E']) >= 0.2
        rev = reverse_positions(d)
        b.set_edge_flags((keep.astype(np.uint8) | (keep[rev].astype(np.uint8) << 1)))
        edge_mask = torch.from_numpy(keep)
    lm = be.dev(lin_mask.astype(np.uint8).reshape(-1))
    grad = be.dev(np.zeros(ws.n_params, np.float32))
    loss = be.dev(np.zeros(2, np.float32))
    ws.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(grad), be.ptr(loss), use_edge_flags=use_dropout,
                 lin_mask=be.ptr(lm), multiply_by=multiply_by, ARR=ARR)
    rl, ro, rg = pyg_ref.loss_and_grads(ref, pyg, ARR=ARR, edge_mask=edge_mask,
                                        lin_mask=torch.from_numpy(lin_mask))
    got_out, got_loss = be.host(out), be.host(loss)
    np.testing.assert_allclose(got_out, ro.numpy(), rtol=rtol, atol=atol)
    assert got_loss[0] == pytest_approx(float(rl), 2e-4)
    gg = unflatten_grads(ws, be.host(grad))
    worst = 0.0
    for key, ref_g in rg.items():
        rgn = ref_g.numpy()
        scale = max(np.abs(rgn).max(), 1e-6)
        err = np.abs(gg[key] - rgn).max() / scale
        worst = max(worst, err)
        assert err < 2e-3, '%s: max rel-to-peak grad error %.3e' % (key, err)
    res.update(train_out=got_out, loss=got_loss, worst_grad_err=worst, ws=ws, batch=b, graph=g, P=P, flat=flat,
               ref=ref, d=d, side=side_buf)
    return res


def pytest_approx(v, rel):
    import pytest
    return pytest.approx(v, rel=rel, abs=1e-6)
