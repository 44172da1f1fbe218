"""Golden vectors of the MODEL half of the path from the REAL dependency the reference pins.

    python tests/golden/make_pyg_golden.py          # needs torch_geometric==1.4.2 (reference README.md:26)

The arithmetic of ``IGMC.forward`` lives in ``torch_geometric==1.4.2`` (``RGCNConv``, ``dropout_adj``; reference call sites
``models.py:6-7, 182-184, 193-202``), an un-vendored dependency that cannot be installed on the build machine of this
repository (no network).  ``oracle/pyg_ref.py`` therefore RESTATES it and is cross-checked only by an
independent fp64 formulation (``tests/test_oracle_independent.py``).  Run this script once on any machine where that PyG
version exists: it feeds fixed, seed-generated inputs through the real ``RGCNConv`` (forward + autograd backward) and the
real ``dropout_adj`` (incl. ``force_undirected``) and writes ``tests/golden/pyg_1_4_2_golden.npz``.  When that file is
present, ``tests/test_oracle_golden.py::test_pyg_ref_against_real_pyg_golden`` compares ``oracle/pyg_ref.py`` with it --
the pin the reference's own test-suite does not provide.  Inputs are regenerated from the seeds below by BOTH sides, so
the file only carries outputs.
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pyg_1_4_2_golden.npz')
CASES = [dict(name='r5', N=40, E=300, fin=4, fout=32, R=5, Bs=4, seed=11),
         dict(name='r5_wide', N=64, E=900, fin=32, fout=32, R=5, Bs=4, seed=12),
         dict(name='r10', N=30, E=200, fin=32, fout=32, R=10, Bs=4, seed=13),
         dict(name='r71', N=50, E=400, fin=32, fout=32, R=71, Bs=4, seed=14)]


def case_inputs(c):
    """Seeded inputs of one RGCNConv case (shared with the consuming test)."""
    g = torch.Generator().manual_seed(c['seed'])
    x = torch.randn(c['N'], c['fin'], generator=g)
    edge_index = torch.randint(0, c['N'], (2, c['E']), generator=g)
    edge_type = torch.randint(0, c['R'], (c['E'],), generator=g)
    b = 1.0 / (c['Bs'] * c['fin']) ** 0.5
    u = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * b          # PyG-1.4.2 `uniform(size, tensor)`
    params = dict(basis=u(c['Bs'], c['fin'], c['fout']), att=u(c['R'], c['Bs']), root=u(c['fin'], c['fout']),
                  bias=u(c['fout']))
    gout = torch.randn(c['N'], c['fout'], generator=g)
    return x, edge_index, edge_type, params, gout


def dropout_inputs(seed=21, N=60, E=500):
    """A simple undirected graph stored in both directions with one relation per edge, like IGMC's subgraphs
    (reference util_functions.py:283-284): no duplicates, so ``coalesce`` only sorts."""
    g = torch.Generator().manual_seed(seed)
    pairs = torch.randint(0, N, (4 * E, 2), generator=g)
    pairs = pairs[pairs[:, 0] < pairs[:, 1]]
    key = torch.unique(pairs[:, 0] * N + pairs[:, 1])[:E // 2]
    i, j = key // N, key % N
    r = torch.randint(0, 5, (len(key),), generator=g)
    edge_index = torch.stack([torch.cat([i, j]), torch.cat([j, i])], 0)
    edge_type = torch.cat([r, r])
    return edge_index, edge_type, N


def main():
    import torch_geometric
    if not torch_geometric.__version__.startswith('1.4.'):
        sys.stderr.write('warning: torch_geometric %s, the reference pins 1.4.2 (RGCNConv changed its parameters and its '
                         'default aggregation in 1.6)\n' % torch_geometric.__version__)
    from torch_geometric.nn import RGCNConv
    from torch_geometric.utils import dropout_adj
    out = dict(pyg_version=np.array(torch_geometric.__version__), torch_version=np.array(torch.__version__))
    for c in CASES:
        x, ei, et, params, gout = case_inputs(c)
        conv = RGCNConv(c['fin'], c['fout'], c['R'], num_bases=c['Bs'])
        with torch.no_grad():
            for k, v in params.items():
                getattr(conv, k).copy_(v)
        x = x.clone().requires_grad_(True)
        y = conv(x, ei, et)
        y.backward(gout)
        out[c['name'] + '/y'] = y.detach().numpy()
        out[c['name'] + '/gx'] = x.grad.numpy()
        for k in params:
            out[c['name'] + '/g_' + k] = getattr(conv, k).grad.numpy()
    # dropout_adj: the RNG stream is torch's, so the draw is replayed from the recorded mask (pyg_ref takes masks); what is
    # pinned is the STRUCTURE: which edges survive a given mask, and the force_undirected symmetrisation + coalesce
    ei, et, N = dropout_inputs()
    for fu in (False, True):
        torch.manual_seed(31)
        ei2, et2 = dropout_adj(ei, et, p=0.2, force_undirected=fu, num_nodes=N, training=True)
        torch.manual_seed(31)                      # the mask dropout_adj drew (1.4.2: bernoulli over the (filtered) edges)
        n_draw = int((ei[0] < ei[1]).sum()) if fu else ei.size(1)
        mask = torch.bernoulli(torch.full((n_draw,), 0.8)).to(torch.bool)
        tag = 'dropout_fu%d' % int(fu)
        out[tag + '/edge_index'], out[tag + '/edge_type'], out[tag + '/mask'] = ei2.numpy(), et2.numpy(), mask.numpy()
    ei3, et3 = dropout_adj(ei, et, p=0.2, force_undirected=False, num_nodes=N, training=False)
    out['dropout_eval/identity'] = np.array(bool(torch.equal(ei3, ei) and torch.equal(et3, et)))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
