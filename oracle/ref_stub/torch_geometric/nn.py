"""``torch_geometric.nn`` names imported at ``/root/reference/models.py:6`` (TEST INFRASTRUCTURE ONLY).

``RGCNConv`` and ``global_sort_pool`` are ``oracle/pyg_ref``'s restatements of the PyG-1.4.2 operators (unpinned:
the package is absent); ``GCNConv`` / ``global_add_pool`` exist as names only -- ``GNN.__init__`` builds ``GCNConv``
layers that ``IGMC`` / ``DGCNN_RS`` replace at once (``/root/reference/models.py:21-24,177-184``).
"""
import torch

from oracle import pyg_ref


class RGCNConv(pyg_ref.RGCNConvRef):
    """PyG-1.4.2 signature ``RGCNConv(in_channels, out_channels, num_relations, num_bases)`` and
    ``forward(x, edge_index, edge_type)``; the literal per-edge formulation (``index_select`` + ``bmm`` +
    ``scatter_add``)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases, **kwargs):
        super().__init__(in_channels, out_channels, num_relations, num_bases)

    def forward(self, x, edge_index, edge_type, edge_norm=None, size=None):
        assert edge_norm is None
        return pyg_ref.rgcn_conv(x, edge_index, edge_type, self.basis, self.att, self.root, self.bias)


class GCNConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, **kwargs):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels

    def reset_parameters(self):
        pass

    def forward(self, *a, **k):
        raise NotImplementedError('GCNConv is a name-only stand-in (dead code in the reference: Main.py:364)')


global_sort_pool = pyg_ref.global_sort_pool


def global_add_pool(x, batch, size=None):
    raise NotImplementedError('global_add_pool is a name-only stand-in (GNN.forward is dead code in the reference)')
