"""Second, independently derived formulation of the model half of the oracle.

``oracle/pyg_ref.py`` restates PyG-1.4.2's RGCNConv the way PyG computes it (per-edge composed weight, ``bmm``,
``scatter_add``) and differentiates it with torch autograd; its parity is unpinned because PyG 1.4.2 cannot be
installed here.  This file derives the SAME mathematics a different way -- dense per-relation adjacency matrices,
``H' = tanh(sum_r A_r H W_r + H root + b)`` with ``W_r = sum_b att[r,b] basis_b``, and a hand-written analytic backward
in numpy fp64 (no autograd, no scatter, no per-edge code) -- and requires agreement with ``pyg_ref`` run in fp64 to
1e-10 on outputs, loss (incl. the ARR term of reference train_eval.py:167-174) and every gradient.  Two independent
derivations agreeing to round-off is the strongest pin available on this box; the semantic anchors (aggr='add',
source -> target flow, root weight + bias, no edge_norm, dropout without rescale of the adjacency, MLP dropout scaled
by 2) are the ones SURVEY.md 8(c) lists from the PyG-1.4.2 sources and the reference's call sites
(models.py:190-217).
"""
import numpy as np
import pytest
import torch

from oracle import pyg_ref


def dense_model(params, x_onehot, src, dst, rel, batch, R, lin_keep=None, multiply_by=1.0):
    """Forward + analytic backward of IGMC in dense fp64 numpy.  Returns (out, cache) / gradient closures."""
    N = x_onehot.shape[0]
    A = np.zeros((R, N, N))
    np.add.at(A, (rel, dst, src), 1.0)                     # A_r[i, j] = #edges j -> i of relation r (aggr = add)
    H = [x_onehot.astype(np.float64)]
    W = []
    for l in range(4):
        basis, att, root, bias = (params['convs.%d.%s' % (l, k)] for k in ('basis', 'att', 'root', 'bias'))
        Wl = np.einsum('rb,bio->rio', att, basis)
        W.append(Wl)
        pre = H[-1] @ root + bias
        for r in range(R):
            pre = pre + A[r] @ (H[-1] @ Wl[r])
        H.append(np.tanh(pre))
    states = np.concatenate(H[1:], 1)                      # [N, 128]
    B = int(batch.max()) + 1
    users = np.where(x_onehot[:, 0] == 1)[0]
    items = np.where(x_onehot[:, 1] == 1)[0]
    assert len(users) == B and len(items) == B
    feat = np.concatenate([states[users], states[items]], 1)
    z = feat @ params['lin1.weight'].T + params['lin1.bias']
    a1 = np.maximum(z, 0.0)
    keep = np.ones_like(a1) if lin_keep is None else lin_keep * 2.0      # F.dropout(p=0.5): kept * 1/(1-p)
    a1d = a1 * keep
    out = (a1d @ params['lin2.weight'].T + params['lin2.bias'])[:, 0] * multiply_by

    def backward(gout):
        g = {}
        go = (gout * multiply_by)[:, None]
        g['lin2.weight'] = go.T @ a1d
        g['lin2.bias'] = go.sum(0)
        da1 = (go @ params['lin2.weight']) * keep * (z > 0)
        g['lin1.weight'] = da1.T @ feat
        g['lin1.bias'] = da1.sum(0)
        dfeat = da1 @ params['lin1.weight']
        dstates = np.zeros_like(states)
        np.add.at(dstates, users, dfeat[:, :128])
        np.add.at(dstates, items, dfeat[:, 128:])
        dH = np.zeros((N, 32))
        for l in range(3, -1, -1):
            dH = dH + dstates[:, 32 * l:32 * l + 32]
            dpre = dH * (1.0 - H[l + 1] ** 2)
            basis, att = params['convs.%d.basis' % l], params['convs.%d.att' % l]
            g['convs.%d.bias' % l] = dpre.sum(0)
            g['convs.%d.root' % l] = H[l].T @ dpre
            dX = dpre @ params['convs.%d.root' % l].T
            dW = np.zeros_like(W[l])
            for r in range(R):
                AtD = A[r].T @ dpre                         # messages flow j -> i, gradients i -> j
                dW[r] = H[l].T @ AtD
                dX = dX + AtD @ W[l][r].T
            g['convs.%d.basis' % l] = np.einsum('rb,rio->bio', att, dW)
            g['convs.%d.att' % l] = np.einsum('rio,bio->rb', dW, basis)
            dH = dX
        return g
    return out, backward, W


def arr_value_and_grads(params, R):
    """ARR term of reference train_eval.py:167-174: sum_l sum_r ||W_l[r+1] - W_l[r]||^2 and its gradient."""
    val, g = 0.0, {}
    for l in range(4):
        basis, att = params['convs.%d.basis' % l], params['convs.%d.att' % l]
        Wl = np.einsum('rb,bio->rio', att, basis)
        diff = Wl[1:] - Wl[:-1]
        val += float((diff ** 2).sum())
        dW = np.zeros_like(Wl)
        dW[1:] += 2 * diff
        dW[:-1] -= 2 * diff
        g['convs.%d.basis' % l] = np.einsum('rb,rio->bio', att, dW)
        g['convs.%d.att' % l] = np.einsum('rio,bio->rb', dW, basis)
    return val, g


def random_batch(rng, B, R, L, n_lo=4, n_hi=14):
    """A block-diagonal batch of bipartite 'enclosing subgraphs' with arbitrary (also repeated) typed edges."""
    xs, srcs, dsts, rels, bvec, off = [], [], [], [], [], 0
    for g in range(B):
        nu, nv = int(rng.integers(n_lo, n_hi)), int(rng.integers(n_lo, n_hi))
        lab = np.concatenate([[0], 2 * rng.integers(1, L // 2, nu - 1), [1], 2 * rng.integers(1, L // 2, nv - 1) + 1])
        lab = np.minimum(lab, L - 1)
        x = np.zeros((nu + nv, L))
        x[np.arange(nu + nv), lab] = 1
        m = int(rng.integers(1, nu * nv))
        u, v = rng.integers(0, nu, m), rng.integers(0, nv, m) + nu
        r = rng.integers(0, R, m)
        srcs += [u + off, v + off]
        dsts += [v + off, u + off]
        rels += [r, r]
        xs.append(x)
        bvec.append(np.full(nu + nv, g))
        off += nu + nv
    return (np.concatenate(xs), np.concatenate(srcs), np.concatenate(dsts), np.concatenate(rels), np.concatenate(bvec))


@pytest.mark.parametrize('R,L,seed,mult,arr', [(5, 4, 0, 1.0, 0.001), (10, 4, 1, 1.0, 0.001), (3, 6, 2, 20.0, 0.01),
                                               (71, 4, 3, 1.0, 0.0)])
def test_dense_fp64_formulation_agrees_with_pyg_ref(R, L, seed, mult, arr):
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    B = 6
    x, src, dst, rel, batch = random_batch(rng, B, R, L)
    ref = pyg_ref.IGMCRef(L, (32, 32, 32, 32), R, 4, adj_dropout=0.2, multiply_by=mult, fast=False).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.1 * torch.randn_like(p))

    class PB(object):
        pass
    pb = PB()
    pb.x, pb.edge_index = torch.from_numpy(x), torch.from_numpy(np.stack([src, dst]))
    pb.edge_type, pb.batch = torch.from_numpy(rel), torch.from_numpy(batch)
    pb.y = torch.from_numpy(rng.uniform(1, 5, B))
    pb.num_graphs = B
    edge_keep = rng.random(len(src)) >= 0.2
    lin_keep = rng.random((B, 128)) < 0.5
    params = {k: v.detach().numpy().copy() for k, v in ref.state_dict().items()}
    # ---- eval forward (no dropout at all)
    sse, out_ref = pyg_ref.eval_sse(ref, pb)
    out, _, _ = dense_model(params, x, src, dst, rel, batch, R, multiply_by=mult)
    np.testing.assert_allclose(out, out_ref.numpy(), rtol=0, atol=1e-10 * max(1.0, mult))
    # ---- train step: injected masks, loss = mse + ARR * sum ||W[r+1]-W[r]||^2, every gradient
    loss_ref, o_ref, g_ref = pyg_ref.loss_and_grads(ref, pb, ARR=arr, edge_mask=torch.from_numpy(edge_keep),
                                                    lin_mask=torch.from_numpy(lin_keep))
    out, backward, _ = dense_model(params, x, src[edge_keep], dst[edge_keep], rel[edge_keep], batch, R,
                                   lin_keep=lin_keep.astype(np.float64), multiply_by=mult)
    np.testing.assert_allclose(out, o_ref.numpy(), rtol=0, atol=1e-10 * max(1.0, mult))
    y = pb.y.numpy()
    arr_val, arr_g = arr_value_and_grads(params, R)
    loss = float(((out - y) ** 2).mean()) + arr * arr_val
    assert loss == pytest.approx(float(loss_ref), rel=1e-12, abs=1e-12)
    g = backward(2.0 * (out - y) / B)
    for k, gr in g_ref.items():
        mine = g[k].reshape(gr.shape) + arr * arr_g.get(k, 0.0)
        scale = max(float(gr.abs().max()), 1e-12)
        assert np.abs(mine - gr.numpy()).max() <= 1e-10 * max(scale, 1.0), k


def test_fast_and_reference_message_paths_are_the_same_function():
    """``rgcn_conv`` (the PyG formulation, timed as the CPU baseline) == ``rgcn_conv_fast`` (used for big test cases)."""
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    x, src, dst, rel, batch = random_batch(rng, 4, 5, 4)
    X = torch.randn(x.shape[0], 32, dtype=torch.float64)
    basis, att = torch.randn(4, 32, 32, dtype=torch.float64), torch.randn(5, 4, dtype=torch.float64)
    root, bias = torch.randn(32, 32, dtype=torch.float64), torch.randn(32, dtype=torch.float64)
    ei, et = torch.from_numpy(np.stack([src, dst])), torch.from_numpy(rel)
    a = pyg_ref.rgcn_conv(X, ei, et, basis, att, root, bias)
    b = pyg_ref.rgcn_conv_fast(X, ei, et, basis, att, root, bias)
    assert float((a - b).abs().max()) < 1e-11


def test_force_undirected_dropout_semantics():
    """PyG-1.4.2 dropout_adj(force_undirected=True): ONE draw per undirected edge on the row < col half, both
    directions kept or dropped together, output coalesced (sorted by (row, col))."""
    ei = torch.tensor([[0, 0, 1, 3, 4, 2], [3, 4, 2, 0, 0, 1]])
    et = torch.tensor([0, 4, 2, 0, 4, 2])
    mask = torch.tensor([True, False, True])                # over the row < col half: (0,3) (0,4) (1,2)
    e2, t2 = pyg_ref.dropout_adj(ei, et, p=0.5, force_undirected=True, num_nodes=5, mask=mask)
    assert e2.tolist() == [[0, 1, 2, 3], [3, 2, 1, 0]] and t2.tolist() == [0, 2, 2, 0]
    e3, t3 = pyg_ref.dropout_adj(ei, et, p=0.5, training=False)
    assert e3 is ei and t3 is et
