"""``torch_geometric.utils.dropout_adj`` as imported at ``/root/reference/models.py:7`` (TEST INFRASTRUCTURE ONLY).

The operator is ``oracle/pyg_ref.dropout_adj`` (PyG-1.4.2 restated, unpinned).  ``MASK_SOURCE`` lets a golden-vector
script supply the Bernoulli keep mask (a callable ``n_candidates -> bool tensor``) so that the draw can be recorded
and replayed on the engine; without it the mask is drawn from torch's global generator like the original.
"""
from oracle import pyg_ref

MASK_SOURCE = None


def dropout_adj(edge_index, edge_attr=None, p=0.5, force_undirected=False, num_nodes=None, training=True):
    if not training:
        return edge_index, edge_attr
    mask = None
    if MASK_SOURCE is not None:
        n = int((edge_index[0] < edge_index[1]).sum()) if force_undirected else int(edge_index.shape[1])
        mask = MASK_SOURCE(n)
    return pyg_ref.dropout_adj(edge_index, edge_attr, p=p, force_undirected=force_undirected, num_nodes=num_nodes,
                               training=True, mask=mask)
