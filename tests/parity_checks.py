"""Parity checks shared by the kernel-logic tests (host emulation of the HIP sources, CPU) and the
GPU tests (real gfx950 library through the same C ABI).  A ``Backend`` hides where buffers live."""
import ctypes

import os

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


def extract_case(be, case, replay, max_graphs=None, seed=0, epoch=0, lean=False):
    """Run the engine extraction on a golden case; returns (graph, batch, downloaded dict)."""
    A = case['A']
    g = engine.Graph(A, device=be.device, lib=be.lib)
    B = len(case['recs'])
    b = engine.Batch(g, max_graphs=max_graphs or B, hop=case['h'], max_nodes_per_hop=case['mnph'])
    if lean:
        b.set_lean(True)      # dense blocks only; the CSR is emitted when something asks for it (download)
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
    ranks = []          # normalised id-ranks of the chosen candidates wherever a cap / ratio cut a hop-1 candidate set
    for g, rec in enumerate(case['recs']):
        users, items, ulab, vlab, edges = graph_canonical(d, g)
        i, j = case['links'][g]
        if rec is not None:
            assert len(users) == len(rec['u_nodes']) or case['h'] > 1
            assert len(items) == len(rec['v_nodes']) or case['h'] > 1
        if case['h'] == 1:
            cand_u = set(Acsc.indices[Acsc.indptr[j]:Acsc.indptr[j + 1]].tolist()) - {i}
            cand_v = set(A.indices[A.indptr[i]:A.indptr[i + 1]].tolist()) - {j}
            assert set(users[1:].tolist()) <= cand_u and set(items[1:].tolist()) <= cand_v
            for chosen, cand in ((users[1:], cand_u), (items[1:], cand_v)):
                if 0 < len(chosen) < len(cand):
                    order = {c: r for r, c in enumerate(sorted(cand))}
                    n, k = len(cand), len(chosen)
                    m = np.mean([(order[int(c)] + 0.5) / n for c in chosen])
                    # (mean rank of a simple random k-sample of n equally spaced ranks: 1/2, variance (n^2 - 1) / (12 n^2) (n - k) / (k (n - 1)))
                    ranks.append((m, (n * n - 1.0) / (12.0 * n * n) * (n - k) / (k * (n - 1.0))))
            if rec is None:
                # no reference record (full-size cases): the sizes the reference would produce follow from the candidate
                # sets alone (util_functions.py:222-229: int(ratio * len), then the per-hop cap)
                def size(c):
                    n = len(c)
                    if case['sample_ratio'] < 1.0:
                        n = int(case['sample_ratio'] * n)
                    if case['mnph'] is not None and case['mnph'] < n:
                        n = case['mnph']
                    return n
                assert len(users) == 1 + size(cand_u) and len(items) == 1 + size(cand_v)
        udist = [ulab[int(u)] // 2 for u in users]
        vdist = [vlab[int(v)] // 2 for v in items]
        out = X.subgraph_extraction_labeling((i, j), A, Acsc, case['h'], 1.0, None, None, None,
                                             case['class_values'], case['link_labels'][g],
                                             node_lists=(users, items, udist, vdist))
        ce = X.canonical_edges(users, items, out[0], out[1], out[2])
        assert np.array_equal(ce, edges)
    # The reference draws a UNIFORM subset (random.sample, util_functions.py:222-229): the chosen candidates' id-ranks average
    # 1/2.  A sampler with an id bias -- the k lowest or highest ids, a prefix of the CSR row -- fails here (the distribution
    # proper -- inclusion counts, pairs, sides, epochs -- is tests/test_sampler_stats.py).
    if len(ranks) >= 8:
        mean = float(np.mean([m for m, _ in ranks]))
        sd = float(np.sqrt(np.sum([v for _, v in ranks]))) / len(ranks)
        assert abs(mean - 0.5) < 6.0 * sd, 'sampled candidates are not uniform over the candidate ids: mean id-rank %.3f (sd %.3f, %d sets)' % (mean, sd, len(ranks))


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


# Tolerances of the fp32 model parity (engine vs oracle/pyg_ref on identical subgraphs, weights and masks), set from what
# the MI355X shows (profiles/r03_parity_observed.txt lists the worst case of every GPU test): <= 10x the observed worst.
# Observed worst on the GPU: outputs 2.0e-6, loss 2.9e-7, gradients 4.6e-6 (round 2 asserted 2e-4 / 2e-4 / 2e-3).
OUT_TOL = 2e-5                       # outputs, max error relative to the batch's peak |output| (rating units)
LOSS_RTOL = 3e-6
GRAD_TOL = 5e-5                      # every gradient tensor, max error relative to the tensor's peak


def run_model_parity(be, case, R, ARR=0.001, use_dropout=True, multiply_by=1.0, seed=3, check_eval=True,
                     rtol=None, atol=None, n_side=0, lean=False):
    """Engine forward / loss+grad on one extracted batch vs the PyG-1.4.2 restatement (oracle/pyg_ref.py)
    on IDENTICAL inputs: same subgraphs, same weights, same dropout masks."""
    import torch
    from oracle import pyg_ref
    out_tol = OUT_TOL if rtol is None else rtol
    g, b, d = extract_case(be, case, replay=False, lean=lean)
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
        res['eval_err'] = rel_err(got, ref_out.numpy())
        assert res['eval_err'] < out_tol, 'eval outputs: max error relative to the peak %.3e' % res['eval_err']
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
    gg = unflatten_grads(ws, be.host(grad))
    worst, worst_key = 0.0, ''
    for key, ref_g in rg.items():
        rgn = ref_g.numpy()
        scale = max(np.abs(rgn).max(), 1e-6)
        err = np.abs(gg[key] - rgn).max() / scale
        if err > worst:
            worst, worst_key = err, key
    record_observed('model_parity', R=R, B=int(B), N=int(d['N']), E=int(d['E']), dropout=bool(use_dropout), lean=bool(lean),
                    eval_out_rel=res.get('eval_err'), train_out_rel=rel_err(got_out, ro.numpy()),
                    loss_rel=abs(float(got_loss[0]) - float(rl)) / max(abs(float(rl)), 1e-12), worst_grad_rel=worst,
                    worst_grad_tensor=worst_key)
    assert rel_err(got_out, ro.numpy()) < out_tol, 'train outputs: max error relative to the peak %.3e' % rel_err(got_out, ro.numpy())
    assert got_loss[0] == pytest_approx(float(rl), LOSS_RTOL)
    assert worst < GRAD_TOL, '%s: max rel-to-peak grad error %.3e' % (worst_key, worst)
    res.update(train_out=got_out, loss=got_loss, worst_grad_err=worst, ws=ws, batch=b, graph=g, P=P, flat=flat,
               ref=ref, d=d, side=side_buf, grads=gg, pyg=pyg, lin_mask=lin_mask, edge_keep=keep if use_dropout else None,
               oracle=(float(rl), ro.numpy(), {k: v.numpy() for k, v in rg.items()}))
    return res


def record_observed(kind, **vals):
    """Append the observed worst errors of a parity check to a JSON-lines log (``IGMC_PARITY_LOG``, default
    ``gpurun_out/parity_observed.jsonl`` when that directory exists): the asserted tolerances are set from what the GPU
    actually shows (profiles/r03_parity_observed.txt), not from a guess."""
    import json
    path = os.environ.get('IGMC_PARITY_LOG')
    if path is None:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        if not os.path.isdir(d):
            return
        path = os.path.join(d, 'parity_observed.jsonl')
    rec = dict(kind=kind, test=os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0])
    rec.update({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in vals.items()})
    try:
        with open(path, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass


def rel_err(got, ref):
    """max |got - ref| / max(|ref|_inf, tiny): error relative to the tensor's peak."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12)) if ref.size else 0.0


def pytest_approx(v, rel):
    import pytest
    return pytest.approx(v, rel=rel, abs=1e-6)


# ====================================================================== free-running RNG paths
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def host_edge_keep(seed, step, graph, u, v, direction, p):
    """include/igmc_rng.h: igmc_edge_hash + igmc_u01 >= p, restated on the host."""
    s = _splitmix64((seed ^ 0x45444745) & _M64)
    s = _splitmix64(s ^ step)
    s = _splitmix64(s ^ ((graph << 2) | direction))
    s = _splitmix64(s ^ ((u << 32) | v))
    h = s >> 32
    return np.float32(h >> 8) * np.float32(1.0 / 16777216.0) >= np.float32(p)


def host_unit_keep(seed, step, graph, j):
    """include/igmc_rng.h: igmc_unit_hash + igmc_u01 >= 0.5."""
    s = _splitmix64((seed ^ 0x4D4C5044) & _M64)
    s = _splitmix64(s ^ step)
    s = _splitmix64(s ^ ((graph << 32) | j))
    h = s >> 32
    return np.float32(h >> 8) * np.float32(1.0 / 16777216.0) >= np.float32(0.5)


def check_edge_flags(d, flags, p, force_undirected, seed, step, exact_sample=400):
    """Free-running edge dropout (k_edge_flags; reference models.py:193-198 -> PyG dropout_adj):
    * entry p of the dst-sorted CSR carries bit0 = keep(src -> dst), bit1 = keep(dst -> src): the two entries of an
      undirected edge must agree on both directed draws;
    * keep rate within 3 sigma of 1 - p; without force_undirected the two directions are independent draws
      (|correlation| within 3 sigma of 0); with it they are ONE draw (bit0 == bit1, symmetric);
    * a sample of entries equals the host restatement of the counter-based hash bit for bit."""
    E = d['E']
    assert len(flags) == E and E > 0
    assert np.all(flags <= 3)
    b0, b1 = (flags & 1).astype(np.int64), ((flags >> 1) & 1).astype(np.int64)
    rev = reverse_positions(d)
    assert np.array_equal(b1, b0[rev]), 'the two CSR entries of an edge disagree on a directed keep bit'
    n_draws = E // 2 if force_undirected else E
    rate = b0.mean()
    sigma = np.sqrt(p * (1 - p) / n_draws)
    assert abs(rate - (1 - p)) < 3 * sigma + 1e-12, (rate, 1 - p, sigma)
    if force_undirected:
        assert np.array_equal(b0, b1) and np.array_equal(flags, flags[rev])
    else:
        # each undirected edge once (src < dst): its two directions are independent Bernoulli draws
        N = d['N']
        dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
        half = d['col'].astype(np.int64) < dst
        x, y = b0[half].astype(np.float64), b1[half].astype(np.float64)
        if x.std() > 0 and y.std() > 0 and len(x) > 50:
            corr = np.corrcoef(x, y)[0, 1]
            assert abs(corr) < 3.0 / np.sqrt(len(x)), corr
    # bit-exact against the host restatement of the hash on a sample of entries
    N = d['N']
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
    rng = np.random.default_rng(0)
    for e in rng.choice(E, size=min(exact_sample, E), replace=False):
        i, c = int(dst[e]), int(d['col'][e])
        row_user = d['node_label'][i] % 2 == 0
        gi, gc, gr = int(d['node_gid'][i]), int(d['node_gid'][c]), int(d['node_graph'][i])
        u, v = (gi, gc) if row_user else (gc, gi)
        dir_f = 2 if force_undirected else (1 if row_user else 0)       # col -> row
        dir_t = 2 if force_undirected else (0 if row_user else 1)
        kf = host_edge_keep(seed, step, gr, u, v, dir_f, p)
        kt = host_edge_keep(seed, step, gr, u, v, dir_t, p)
        assert int(flags[e]) == int(kf) | (int(kt) << 1), (e, int(flags[e]), kf, kt)


def run_free_running_dropout(be, case, R, p=0.2, force_undirected=False, seed=11, step=7, mlp_seed=5, mlp_step=3,
                             lean=False):
    """Edge dropout AND MLP dropout drawn by the kernels themselves (no injected masks): the drawn masks are read
    back, checked statistically and against the host restatement of the hashes, and the model's loss / gradients
    with them must equal the oracle's with the same masks -- for force_undirected through the oracle's own
    ``dropout_adj(force_undirected=True)`` (mask over the row < col half, re-symmetrised, coalesced)."""
    import torch
    from oracle import pyg_ref
    g, b, d = extract_case(be, case, replay=False, lean=lean)
    L = 2 * case['h'] + 2
    # (a lean arena draws on the dense blocks -- k_relm_dropout -- and the CSR read back below takes its flags from them)
    b.edge_dropout(p, force_undirected, seed=seed, step=step)
    be.sync()
    flags = b.download()['eflag']
    check_edge_flags(d, flags, p, force_undirected, seed, step)
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    ref = make_ref_model(L, R, seed=3, adj_dropout=p)
    ref.force_undirected = bool(force_undirected)
    P = be.dev(flatten_params(ws, ref))
    B = d['B']
    out, grad, loss = be.dev(np.zeros(B, np.float32)), be.dev(np.zeros(ws.n_params, np.float32)), be.dev(np.zeros(2, np.float32))
    ws.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(grad), be.ptr(loss), use_edge_flags=True, lin_mask=None,
                 seed=mlp_seed, step=mlp_step, ARR=0.001)
    be.sync()
    lm = np.zeros(B * 128, np.uint8)
    be.lib.call('igmc_debug_lin_mask', ws.handle, ctypes.c_void_p(lm.ctypes.data), B * 128)
    lm = lm.reshape(B, 128)
    # MLP dropout(0.5): rate within 3 sigma, bit-exact vs the host restatement of igmc_unit_hash
    assert abs(lm.mean() - 0.5) < 3 * 0.5 / np.sqrt(lm.size), lm.mean()
    for gi in range(0, B, max(1, B // 4)):
        for j in (0, 1, 63, 127):
            assert bool(lm[gi, j]) == bool(host_unit_keep(mlp_seed, mlp_step, gi, j))
    pyg = batch_to_pyg(d, L)
    keep = torch.from_numpy((flags & 1).astype(bool))
    if force_undirected:
        sel = (pyg.edge_index[0] < pyg.edge_index[1]).numpy()      # the half dropout_adj draws on
        keep = keep[torch.from_numpy(sel)]
    rl, ro, rg = pyg_ref.loss_and_grads(ref, pyg, ARR=0.001, edge_mask=keep, lin_mask=torch.from_numpy(lm.astype(bool)))
    gg = unflatten_grads(ws, be.host(grad))
    worst, worst_key = 0.0, ''
    for key, ref_g in rg.items():
        rgn = ref_g.numpy()
        err = np.abs(gg[key] - rgn).max() / max(np.abs(rgn).max(), 1e-6)
        if err > worst:
            worst, worst_key = err, key
    record_observed('free_running_dropout', R=R, B=int(B), N=int(d['N']), E=int(d['E']), lean=bool(lean),
                    force_undirected=bool(force_undirected), train_out_rel=rel_err(be.host(out), ro.numpy()),
                    loss_rel=abs(float(be.host(loss)[0]) - float(rl)) / max(abs(float(rl)), 1e-12), worst_grad_rel=worst,
                    worst_grad_tensor=worst_key)
    assert rel_err(be.host(out), ro.numpy()) < OUT_TOL
    assert be.host(loss)[0] == pytest_approx(float(rl), LOSS_RTOL)
    assert worst < GRAD_TOL, '%s: max rel-to-peak grad error %.3e' % (worst_key, worst)
    return dict(worst_grad_err=worst, keep_rate=float((flags & 1).mean()), lin_rate=float(lm.mean()))


# fused multi-step trajectory vs pyg_ref.train_step + torch.optim.Adam (same provenance as the tolerances above)
# Observed worst on the GPU (5-10 steps): loss 2.9e-7, exp_avg 1.2e-5, exp_avg_sq 3.4e-5, 2e-5 of the parameters further than
# 2e-5 + 2e-3 |p| from the oracle's, largest parameter difference 3.1e-5 (round 2 asserted 5e-4 / 2e-3 / 4e-3 / 2e-3 / 2 lr steps)
TRAJ_LOSS_RTOL = 3e-6
TRAJ_M1_TOL, TRAJ_M2_TOL = 1.2e-4, 3.5e-4      # Adam moments, relative to the tensor's peak
TRAJ_P_ATOL, TRAJ_P_RTOL, TRAJ_FRAC_OFF = 2e-5, 2e-3, 2e-4
TRAJ_P_MAX = 3e-4                              # largest parameter difference after the steps


def run_fused_train_trajectory(be, case, R, steps=5, batch=None, use_dropout=False, lr=1e-3, ARR=0.001, seed=4):
    """The trajectory check below with the ORACLE on one torch thread: the order of its CPU scatter-adds then does not depend
    on the host (128 / 8 / 1 threads moved flixster's exp_avg deviation between 8e-6 and 1.3e-4 with the engine's bits
    unchanged -- one parameter whose gradient is within float noise of zero takes its Adam step the other way:
    profiles/r04_flixster_trajectory_oracle_spread.txt)."""
    import torch
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _run_fused_train_trajectory(be, case, R, steps, batch, use_dropout, lr, ARR, seed)
    finally:
        torch.set_num_threads(nt)


def _run_fused_train_trajectory(be, case, R, steps=5, batch=None, use_dropout=False, lr=1e-3, ARR=0.001, seed=4):
    """``igmc_train_step`` (the fused single-GPU step: k_graph_step -> k_tail_ts -> k_finalize_ts with Adam, or the
    per-layer kernels + k_finalize where the subgraph kernel is not eligible) for ``steps`` consecutive steps on
    DIFFERENT batches with injected masks, against ``pyg_ref.train_step`` + ``torch.optim.Adam`` on the same batches
    (reference train_eval.py:158-177).  Returns per-step losses and the final parameter / Adam-state comparison."""
    import torch
    from oracle import pyg_ref
    A = case['A']
    L = 2 * case['h'] + 2
    n = len(case['links'])
    B = batch or n // steps
    assert B * steps <= n
    g = engine.Graph(A, device=be.device, lib=be.lib)
    b = engine.Batch(g, max_graphs=B, hop=case['h'], max_nodes_per_hop=case['mnph'])
    ys = case['class_values'][case['link_labels']].astype(np.float32)
    lu, lv, ly = be.dev(case['links'][:, 0].astype(np.int32)), be.dev(case['links'][:, 1].astype(np.int32)), be.dev(ys)
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, 0, b.node_capacity, b.edge_capacity, B)
    ref = make_ref_model(L, R, seed=seed, adj_dropout=0.2 if use_dropout else 0.0)
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    n_p = ws.n_params
    P = be.dev(flatten_params(ws, ref))
    M1, M2, G = be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32))
    out, loss, total = be.dev(np.zeros(B, np.float32)), be.dev(np.zeros(2, np.float32)), be.dev(np.zeros(1, np.float64))
    rng = np.random.default_rng(seed)
    losses = []
    for s in range(steps):
        b.extract(be.ptr(lu), be.ptr(lv), be.ptr(ly), None, s * B, B, sample_ratio=case['sample_ratio'], seed=2, epoch=1)
        be.sync()
        d = b.download()
        pyg = batch_to_pyg(d, L)
        lm = rng.random((B, 128)) < 0.5
        LM = be.dev(lm.astype(np.uint8).reshape(-1))
        edge_mask = None
        if use_dropout:
            keep = rng.random(d['E']) >= 0.2
            rev = reverse_positions(d)
            b.set_edge_flags(keep.astype(np.uint8) | (keep[rev].astype(np.uint8) << 1))
            edge_mask = torch.from_numpy(keep)
        be.lib.call('igmc_train_step', ws.handle, engine._p(be.ptr(P)), b.handle, int(use_dropout), engine._p(be.ptr(LM)),
                    0, 0, 1.0, ARR, engine._p(be.ptr(out)), engine._p(be.ptr(G)), engine._p(be.ptr(M1)),
                    engine._p(be.ptr(M2)), engine._p(be.ptr(loss)), engine._p(be.ptr(total)), None, s + 1, lr, 0.9, 0.999,
                    1e-8, 0.0, None)
        be.sync()
        ref_loss = pyg_ref.train_step(ref, opt, pyg, ARR=ARR, edge_mask=edge_mask, lin_mask=torch.from_numpy(lm))
        got = float(be.host(loss)[0])
        losses.append((got, ref_loss))
    be.lib.call('igmc_model_check', ws.handle, None)
    # Adam state: exp_avg is LINEAR in the gradients -> tight, per tensor relative to its peak
    m1 = unflatten_grads(ws, be.host(M1))
    m2 = unflatten_grads(ws, be.host(M2))
    worst_m1 = worst_m2 = 0.0
    for i, (k, prm) in enumerate(ref.named_parameters()):
        st = opt.state[prm]
        ea, es = st['exp_avg'].numpy(), st['exp_avg_sq'].numpy()
        worst_m1 = max(worst_m1, float(np.abs(m1[k] - ea).max() / max(np.abs(ea).max(), 1e-9)))
        worst_m2 = max(worst_m2, float(np.abs(m2[k] - es).max() / max(np.abs(es).max(), 1e-12)))
    got_p, want_p = be.host(P), flatten_params(ws, ref)
    diff = np.abs(got_p - want_p)
    tol = TRAJ_P_ATOL + TRAJ_P_RTOL * np.abs(want_p)
    # Adam divides by sqrt(v): an element whose gradient is within float noise of zero may take its (<= lr) step in
    # the other direction, so a handful of elements can be off by up to 2 * lr * steps; everything else must track
    bad = diff > tol
    loss_rel = max(abs(a - b) / max(abs(b), 1e-12) for a, b in losses)
    record_observed('fused_trajectory', R=R, steps=steps, batch=int(B), dropout=bool(use_dropout), loss_rel=loss_rel,
                    exp_avg_rel=worst_m1, exp_avg_sq_rel=worst_m2, params_frac_off=float(bad.mean()),
                    params_max_diff=float(diff.max()), lr_steps=lr * steps)
    assert loss_rel < TRAJ_LOSS_RTOL, losses
    assert worst_m1 <= TRAJ_M1_TOL and worst_m2 <= TRAJ_M2_TOL, (worst_m1, worst_m2)
    assert bad.mean() < TRAJ_FRAC_OFF, 'too many parameters off the oracle trajectory: %g' % bad.mean()
    assert diff.max() <= TRAJ_P_MAX, diff.max()
    return dict(losses=losses, frac_off=float(bad.mean()), max_diff=float(diff.max()), total=float(be.host(total)[0]),
                params=got_p, m1=be.host(M1), m2=be.host(M2), ws=ws)


# ====================================================================== DGCNN_RS (sort-pool readout family)
# (sort-pool family: <= 10 x the worst the GPU shows over its cases -- outputs 1.2e-6, loss 1.4e-7, gradients 2.2e-6 of the
#  tensor's peak; profiles/r03_parity_observed.txt)
DGCNN_OUT_TOL, DGCNN_LOSS_RTOL, DGCNN_GRAD_TOL = 1.2e-5, 1.5e-6, 2.2e-5


def run_dgcnn_parity(be, case, R, k=12, use_dropout=True, ARR=0.001, seed=3, rtol=None, atol=None):
    """``igmc_sortpool_forward`` / ``igmc_sortpool_loss_grad`` (conv kernels of model.hip + sortpool.hip) on one
    extracted batch vs ``pyg_ref.DGCNNRSRef`` (reference models.py:123-167 restated) on IDENTICAL subgraphs, weights and
    dropout masks: eval outputs, training outputs, loss and every gradient."""
    import torch
    from oracle import pyg_ref
    g, b, d = extract_case(be, case, replay=False)
    L = 2 * case['h'] + 2
    B = d['B']
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    sp = engine.SortPoolWorkspace(ws, k, max(2, b.node_capacity // b.max_graphs))
    torch.manual_seed(seed)
    ref = pyg_ref.DGCNNRSRef(L, (32, 32, 32, 1), k, R, 4, adj_dropout=0.2 if use_dropout else 0.0, fast=True)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    sd = {key: v.detach().cpu().numpy().astype(np.float32) for key, v in ref.state_dict().items()}
    flat = np.zeros(sp.n_params, np.float32)
    for key, off in sp.offsets.items():
        a = sd[key].reshape(-1)
        flat[off:off + a.size] = a
    assert sum(v.size for v in sd.values()) == sp.n_params
    P = be.dev(flat)
    out = be.dev(np.zeros(B, np.float32))
    pyg = batch_to_pyg(d, L)
    # ---- evaluation forward
    sp.forward(be.ptr(P), b, be.ptr(out), training=False)
    ref.eval()
    with torch.no_grad():
        ro = ref(pyg)
    eval_rel = rel_err(be.host(out), ro.numpy())
    assert eval_rel < DGCNN_OUT_TOL, eval_rel
    # ---- training step with injected masks
    rng = np.random.default_rng(seed)
    lin_mask = rng.random((B, 128)) < 0.5
    edge_mask = None
    if use_dropout:
        keep = rng.random(d['E']) >= 0.2
        rev = reverse_positions(d)
        b.set_edge_flags((keep.astype(np.uint8) | (keep[rev].astype(np.uint8) << 1)))
        edge_mask = torch.from_numpy(keep)
    lm = be.dev(lin_mask.astype(np.uint8).reshape(-1))
    grad = be.dev(np.zeros(sp.n_params, np.float32))
    loss = be.dev(np.zeros(2, np.float32))
    sp.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(grad), be.ptr(loss), use_edge_flags=use_dropout, lin_mask=be.ptr(lm), ARR=ARR)
    ref.train()
    ref.zero_grad()
    to = ref(pyg, edge_mask=edge_mask, lin_mask=torch.from_numpy(lin_mask))
    rl = F_mse(to, pyg.y) + ARR * pyg_ref.arr_loss(ref)
    rl.backward()
    gflat = be.host(grad)
    worst, worst_key = 0.0, ''
    for key, p in ref.named_parameters():
        rgn = p.grad.detach().numpy()
        off = sp.offsets[key]
        got = gflat[off:off + rgn.size].reshape(rgn.shape)
        scale = max(np.abs(rgn).max(), 1e-6)
        err = np.abs(got - rgn).max() / scale
        if err > worst:
            worst, worst_key = err, key
    train_rel = rel_err(be.host(out), to.detach().numpy())
    loss_rel = abs(float(be.host(loss)[0]) - float(rl.detach())) / max(abs(float(rl.detach())), 1e-12)
    record_observed('dgcnn_parity', R=R, B=int(B), k=int(k), dropout=bool(use_dropout), eval_out_rel=eval_rel,
                    train_out_rel=train_rel, loss_rel=loss_rel, worst_grad_rel=worst, worst_grad_tensor=worst_key)
    assert train_rel < DGCNN_OUT_TOL, train_rel
    assert loss_rel < DGCNN_LOSS_RTOL, loss_rel
    assert worst < DGCNN_GRAD_TOL, '%s: max rel-to-peak grad error %.3e' % (worst_key, worst)
    return dict(worst_grad_err=worst, eval_out=be.host(out), loss=be.host(loss))


def F_mse(a, b):
    import torch
    return torch.nn.functional.mse_loss(a, b.view(-1).to(a.dtype))


# ====================================================================== fixtures of the reference's own models.py / train_eval.py
def fixture_case(mg):
    """Extraction case (graph, links, recorded node lists) of a ``tests/golden/model_golden.npz`` entry.  The graph is
    rebuilt here -- synthetic graphs are deterministic, flixster is bundled -- and its fingerprint compared."""
    import hashlib
    import scipy.sparse as ssp
    from igmc_amd import preprocessing
    name = mg.case
    if name.startswith('headline'):
        A = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)[2]
        mnph = 100
    elif name in ('igmc_r5', 'igmc_side'):
        u, v, r = preprocessing.synth_ml(300, 200, 9000, preprocessing.ML_HIST['ml_100k'][3], seed=3)
        A, mnph = ssp.csr_matrix((r.astype(np.float32), (u, v)), shape=(300, 200)), 15
    else:
        A, mnph = preprocessing.load_data_monti('flixster', testing=True)[2], 10000
    S = ssp.csr_matrix(A)
    S.sort_indices()
    h = hashlib.sha256()
    for a in (S.indptr.astype(np.int64), S.indices.astype(np.int64), S.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert np.frombuffer(h.digest()[:8], np.uint64)[0] == mg['graph_fingerprint'], 'not the graph of the fixture'
    uo, vo = mg['u_off'], mg['v_off']
    recs = []
    for g in range(len(uo) - 1):
        un, vn = mg['u_nodes'][uo[g]:uo[g + 1]].astype(np.int64), mg['v_nodes'][vo[g]:vo[g + 1]].astype(np.int64)
        labels = np.concatenate([np.where(np.arange(len(un)) == 0, 0, 2), np.where(np.arange(len(vn)) == 0, 1, 3)])
        recs.append(dict(u_nodes=un, v_nodes=vn, labels=labels.astype(np.int64)))
    return dict(A=A, links=mg['links'].astype(np.int64), link_labels=mg['link_labels'].astype(np.int64),
                class_values=mg['class_values'], h=1, sample_ratio=1.0, mnph=mnph, recs=recs)


def reference_edges(case, first, B):
    """Directed edges of graphs ``first .. first+B`` in the REFERENCE's collated order (construct_pyg_graph:
    ``[u; v]`` then ``[v; u]`` per graph, util_functions.py:283) as keys ``(graph, side of src, src gid, dst gid)``;
    rebuilt from the recorded node lists by the pinned extraction oracle."""
    from oracle import extract_ref as X
    A, Acsc = case['A'], case['A'].tocsc()
    keys = []
    for g in range(B):
        rec = case['recs'][first + g]
        un, vn = rec['u_nodes'], rec['v_nodes']
        nu = len(un)
        out = X.subgraph_extraction_labeling(tuple(case['links'][first + g]), A, Acsc, 1, 1.0, None, None, None,
                                             case['class_values'], case['link_labels'][first + g],
                                             node_lists=(un, vn, [0] + [1] * (nu - 1), [0] + [1] * (len(vn) - 1)))
        u, v = out[0], out[1] - nu
        keys += [(g, 0, int(un[a]), int(vn[b])) for a, b in zip(u, v)]       # user -> item
        keys += [(g, 1, int(vn[b]), int(un[a])) for a, b in zip(u, v)]       # item -> user
    return keys


def engine_edge_flags(d, ref_keys, keep, force_undirected):
    """Keep flags per entry of the engine's dst-sorted CSR (bit 0: src -> dst kept, bit 1: dst -> src kept) from a keep
    mask drawn over the REFERENCE's edge order (all directed edges, or the ``row < col`` = user -> item half under
    ``force_undirected``: PyG ``dropout_adj`` as called at reference models.py:193-198)."""
    if force_undirected:
        half = [k for k in ref_keys if k[1] == 0]
        assert len(half) == len(keep)
        kept = {(g, s, t) for (g, _, s, t), kp in zip(half, keep) if kp}
        look = lambda g, side, s, t: (g, s, t) in kept if side == 0 else (g, t, s) in kept
        lookr = look
    else:
        assert len(ref_keys) == len(keep)
        kd = {k: bool(kp) for k, kp in zip(ref_keys, keep)}
        look = lambda g, side, s, t: kd[(g, side, s, t)]
    N = d['N']
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
    flags = np.zeros(d['E'], np.uint8)
    for p in range(d['E']):
        i, c = int(dst[p]), int(d['col'][p])
        g = int(d['node_graph'][i])
        side_src = int(d['node_label'][c] % 2)
        s, t = int(d['node_gid'][c]), int(d['node_gid'][i])
        fwd = look(g, side_src, s, t)
        bwd = look(g, 1 - side_src, t, s)
        flags[p] = int(fwd) | (int(bwd) << 1)
    return flags


def run_reference_fixture(be, mg, batch_size, tol_scale=1.0, trajectory=True):
    """The engine on the SAME subgraphs, weights and dropout masks as the reference's own ``models.py`` /
    ``train_eval.py`` run recorded in ``tests/golden/model_golden.npz``: eval outputs, first-step outputs and every
    gradient, then the fused ``igmc_train_step`` over the epoch's batches against the parameters the reference's
    ``train`` + ``torch.optim.Adam`` left, and its returned epoch loss."""
    case = fixture_case(mg)
    R, n_side, L = mg.R, mg.n_side, 4
    n_batches = mg.n_steps()
    g = engine.Graph(case['A'], device=be.device, lib=be.lib)
    b = engine.Batch(g, max_graphs=batch_size, hop=1, max_nodes_per_hop=case['mnph'])
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, n_side, b.node_capacity, b.edge_capacity, b.max_graphs)
    init = {k: v.numpy() for k, v in mg.state('init').items()}
    flat = np.zeros(ws.n_params, np.float32)
    for key, off, shape in ws.layout():
        flat[off:off + int(np.prod(shape))] = init[key].reshape(-1)
    P = be.dev(flat)
    n_p = ws.n_params
    M1, M2, G = be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32)), be.dev(np.zeros(n_p, np.float32))
    out, loss = be.dev(np.zeros(batch_size, np.float32)), be.dev(np.zeros(2, np.float32))
    total = be.dev(np.zeros(1, np.float64))
    ys_all = case['class_values'][case['link_labels']].astype(np.float32)
    res = dict(eval_rel=0.0)
    side_bufs = []

    def load_batch(s):
        first = s * batch_size
        recs = case['recs'][first:first + batch_size]
        ul, vl = [r['u_nodes'] for r in recs], [r['v_nodes'] for r in recs]
        ud = [np.r_[0, np.ones(len(u) - 1, np.int64)] for u in ul]
        vd = [np.r_[0, np.ones(len(v) - 1, np.int64)] for v in vl]
        b.extract_replay(ul, vl, ud, vd, ys_all[first:first + batch_size])
        be.sync()
        d = b.download()
        if n_side:
            uf = mg['u_features'][case['links'][first:first + batch_size, 0]]
            vf = mg['v_features'][case['links'][first:first + batch_size, 1]]
            sb = be.dev(np.concatenate([uf, vf], 1).astype(np.float32))
            side_bufs.append(sb)
            b.set_side_features(be.ptr(sb), n_side)
        flags = None
        em = mg.edge_mask(s)
        if em is not None:
            flags = engine_edge_flags(d, reference_edges(case, first, len(recs)), em, mg.force_undirected)
            b.set_edge_flags(flags)
        lm = be.dev(mg.lin_mask(s, (len(recs), 128)).astype(np.uint8).reshape(-1))
        return d, flags is not None, lm

    # ---- eval-mode forward of every batch with the initial parameters
    for s in range(n_batches):
        d, _, _ = load_batch(s)
        ws.forward(be.ptr(P), b, be.ptr(out), training=False, multiply_by=mg.multiply_by)
        res['eval_rel'] = max(res['eval_rel'], rel_err(be.host(out)[:d['B']], mg['eval_out/%d' % s]))
    assert res['eval_rel'] < OUT_TOL * tol_scale, res['eval_rel']
    # ---- first step: outputs + every gradient
    d, use_flags, lm = load_batch(0)
    grad = be.dev(np.zeros(n_p, np.float32))
    ws.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(grad), be.ptr(loss), use_edge_flags=use_flags, lin_mask=be.ptr(lm),
                 multiply_by=mg.multiply_by, ARR=mg.ARR)
    res['train_out_rel'] = rel_err(be.host(out)[:d['B']], mg['train/out/0'])
    gg = unflatten_grads(ws, be.host(grad))
    ref_g = {k: v.numpy() for k, v in mg.state('train/grad0').items()}
    worst, worst_key = 0.0, ''
    for key, rgn in ref_g.items():
        err = np.abs(gg[key] - rgn).max() / max(np.abs(rgn).max(), 1e-6)
        if err > worst:
            worst, worst_key = float(err), key
    res.update(worst_grad_rel=worst, worst_grad_tensor=worst_key, N=int(d['N']), E=int(d['E']))
    if n_batches == 1:          # one batch: the reference's returned epoch loss IS the step's loss
        res['loss_rel'] = abs(float(be.host(loss)[0]) - float(mg['train/epoch_loss'])) / abs(float(mg['train/epoch_loss']))
        assert res['loss_rel'] < LOSS_RTOL * tol_scale, res['loss_rel']
    assert res['train_out_rel'] < OUT_TOL * tol_scale, res['train_out_rel']
    assert worst < GRAD_TOL * tol_scale, '%s: max rel-to-peak grad error %.3e' % (worst_key, worst)
    # ---- the epoch through the fused step (loss bookkeeping + Adam inside the kernels) vs the reference's train()
    if trajectory:
        tot = 0.0
        for s in range(n_batches):
            d, use_flags, lm = load_batch(s)
            be.lib.call('igmc_train_step', ws.handle, engine._p(be.ptr(P)), b.handle, int(use_flags), engine._p(be.ptr(lm)),
                        0, 0, float(mg.multiply_by), mg.ARR, engine._p(be.ptr(out)), engine._p(be.ptr(G)),
                        engine._p(be.ptr(M1)), engine._p(be.ptr(M2)), engine._p(be.ptr(loss)), engine._p(be.ptr(total)), None,
                        s + 1, mg.lr, 0.9, 0.999, 1e-8, 0.0, None)
            be.sync()
            tot += float(be.host(loss)[0]) * d['B']
            assert rel_err(be.host(out)[:d['B']], mg['train/out/%d' % s]) < 5 * OUT_TOL * tol_scale
        be.lib.call('igmc_model_check', ws.handle, None)
        n = n_batches * batch_size
        res['epoch_loss_rel'] = abs(tot / n - float(mg['train/epoch_loss'])) / abs(float(mg['train/epoch_loss']))
        assert res['epoch_loss_rel'] < TRAJ_LOSS_RTOL * tol_scale, res['epoch_loss_rel']
        assert float(be.host(total)[0]) == pytest_approx(tot, 1e-5)
        post = {k: v.numpy() for k, v in mg.state('post').items()}
        want = np.zeros(n_p, np.float32)
        for key, off, shape in ws.layout():
            want[off:off + int(np.prod(shape))] = post[key].reshape(-1)
        diff = np.abs(be.host(P) - want)
        bad = diff > TRAJ_P_ATOL + TRAJ_P_RTOL * np.abs(want)
        res.update(params_frac_off=float(bad.mean()), params_max_diff=float(diff.max()))
        assert bad.mean() < TRAJ_FRAC_OFF and diff.max() <= TRAJ_P_MAX, (bad.mean(), diff.max())
    record_observed('reference_fixture', case=mg.case, **{k: v for k, v in res.items()})
    return res


def run_reference_fixture_dgcnn(be, mg, batch_size):
    """Sort-pool family: the engine (conv kernels + sortpool.hip) on the subgraphs / weights / masks of the reference's own
    ``DGCNN_RS`` run (``models.py:123-167``): eval outputs of every batch, first-step outputs and every gradient."""
    case = fixture_case(mg)
    R, L, k = mg.R, 4, int(mg['k'])
    g = engine.Graph(case['A'], device=be.device, lib=be.lib)
    b = engine.Batch(g, max_graphs=batch_size, hop=1, max_nodes_per_hop=case['mnph'])
    ws = engine.ModelWorkspace(be.lib, be.device, R, 4, L, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    sp = engine.SortPoolWorkspace(ws, k, max(2, b.node_capacity // b.max_graphs))
    init = {key: v.numpy() for key, v in mg.state('init').items()}
    flat = np.zeros(sp.n_params, np.float32)
    for key, off in sp.offsets.items():
        flat[off:off + init[key].size] = init[key].reshape(-1)
    assert sum(v.size for v in init.values()) == sp.n_params
    P = be.dev(flat)
    out, loss = be.dev(np.zeros(batch_size, np.float32)), be.dev(np.zeros(2, np.float32))
    ys_all = case['class_values'][case['link_labels']].astype(np.float32)

    def load_batch(s):
        first = s * batch_size
        recs = case['recs'][first:first + batch_size]
        ul, vl = [r['u_nodes'] for r in recs], [r['v_nodes'] for r in recs]
        b.extract_replay(ul, vl, [np.r_[0, np.ones(len(u) - 1, np.int64)] for u in ul],
                         [np.r_[0, np.ones(len(v) - 1, np.int64)] for v in vl], ys_all[first:first + batch_size])
        be.sync()
        return b.download(), first, len(recs)

    res = dict(eval_rel=0.0)
    for s in range(mg.n_steps()):
        d, _, _ = load_batch(s)
        sp.forward(be.ptr(P), b, be.ptr(out), training=False)
        res['eval_rel'] = max(res['eval_rel'], rel_err(be.host(out)[:d['B']], mg['eval_out/%d' % s]))
    assert res['eval_rel'] < DGCNN_OUT_TOL, res['eval_rel']
    d, first, nb = load_batch(0)
    b.set_edge_flags(engine_edge_flags(d, reference_edges(case, first, nb), mg.edge_mask(0), mg.force_undirected))
    lm = be.dev(mg.lin_mask(0, (nb, 128)).astype(np.uint8).reshape(-1))
    grad = be.dev(np.zeros(sp.n_params, np.float32))
    sp.loss_grad(be.ptr(P), b, be.ptr(out), be.ptr(grad), be.ptr(loss), use_edge_flags=True, lin_mask=be.ptr(lm), ARR=mg.ARR)
    res['train_out_rel'] = rel_err(be.host(out)[:d['B']], mg['train/out/0'])
    gflat = be.host(grad)
    worst, worst_key = 0.0, ''
    for key, v in mg.state('train/grad0').items():
        rgn = v.numpy()
        off = sp.offsets[key]
        err = np.abs(gflat[off:off + rgn.size].reshape(rgn.shape) - rgn).max() / max(np.abs(rgn).max(), 1e-6)
        if err > worst:
            worst, worst_key = float(err), key
    res.update(worst_grad_rel=worst, worst_grad_tensor=worst_key, k=k)
    record_observed('reference_fixture', case=mg.case, **res)
    assert res['train_out_rel'] < DGCNN_OUT_TOL, res['train_out_rel']
    assert worst < DGCNN_GRAD_TOL, '%s: max rel-to-peak grad error %.3e' % (worst_key, worst)
    return res
