"""Pure-torch CPU restatement of the PyG-1.4.2 operators behind ``models.IGMC`` and of the model / train / eval code
around them (TEST ORACLE).

MODEL CODE PINNED to the reference's own files; THREE OPERATORS UNPINNED:

* ``IGMCRef.forward`` (``/root/reference/models.py:190-217``), ``DGCNNRSRef`` (``:123-167`` on ``:63-120``),
  ``train_step`` / ``loss_and_grads`` (``/root/reference/train_eval.py:157-177``), ``eval_sse`` (``:182-199``) and the
  ensemble mean (``:208-239``) are checked by ``tests/test_model_golden.py`` against golden vectors produced by the
  UNMODIFIED reference ``models.py`` / ``train_eval.py`` (``tests/golden/make_model_golden.py``: imported through the
  ``torch_geometric`` stand-in of ``oracle/ref_stub``) -- outputs, every gradient, the epoch loss of ``train``, the
  parameters after Adam, ``eval_loss`` / ``eval_rmse`` and their ensemble forms, to 1e-6 of the tensor's peak.
* ``rgcn_conv`` (PyG ``RGCNConv``), ``dropout_adj``, ``global_sort_pool`` and the ``Batch`` collate restate the
  published PyG-1.4.2 algorithm: ``torch_geometric==1.4.2``
  (``/root/reference/README.md:26``) is an un-vendored dependency that is absent from ``/root/reference`` and cannot be
  installed here, and the reference holds no tests / golden vectors for it -- PARITY UNPINNED for these; the golden
  generator binds the reference's imports of them to the functions below.  They are anchored on the reference's own
  call sites: ``RGCNConv`` at ``/root/reference/models.py:182-184, 200-202`` (parameter names / shapes and
  ``W = att @ basis.view(num_bases,-1)`` are fixed by the in-tree ARR code ``/root/reference/train_eval.py:167-174``),
  ``dropout_adj`` at ``models.py:193-198``, the collate through ``DataLoader`` at ``train_eval.py:44-51``; on an fp64
  dense-adjacency twin (``tests/test_oracle_independent.py``); and on ``tests/golden/make_pyg_golden.py`` for any
  machine that has the real package.

The RGCNConv message path deliberately keeps the reference formulation
(per-edge ``index_select`` of the composed weight + ``bmm`` + ``scatter_add``) so
that it is also the honest "reference CPU path" timed by ``bench.py``'s
``cpu_baseline`` leg.

This module is test infrastructure: the product package never imports it.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- PyG ops
def pyg_uniform_(size, tensor):
    """``torch_geometric.nn.inits.uniform``: U(-1/sqrt(size), 1/sqrt(size))."""
    bound = 1.0 / math.sqrt(size)
    with torch.no_grad():
        tensor.uniform_(-bound, bound)


def rgcn_conv(x, edge_index, edge_type, basis, att, root, bias):
    """PyG-1.4.2 ``RGCNConv.forward`` with ``aggr='add'``, flow source->target.

    message: ``w = (att @ basis.view(Bs,-1)).view(R,in,out)[edge_type]``;
    ``out = bmm(x_j.unsqueeze(1), w).squeeze(-2)``; aggregate: ``scatter_add`` at
    ``edge_index[1]``; update: ``aggr + x @ root + bias``.
    """
    num_bases, cin, cout = basis.shape
    num_rel = att.shape[0]
    w = torch.matmul(att, basis.view(num_bases, -1)).view(num_rel, cin, cout)
    src, dst = edge_index[0], edge_index[1]
    w_e = torch.index_select(w, 0, edge_type)                       # [E, in, out]
    msg = torch.bmm(x[src].unsqueeze(1), w_e).squeeze(-2)           # [E, out]
    aggr = torch.zeros(x.shape[0], cout, dtype=x.dtype).index_add_(0, dst, msg)
    return aggr + torch.matmul(x, root) + bias


def rgcn_conv_fast(x, edge_index, edge_type, basis, att, root, bias):
    """Same math as :func:`rgcn_conv`, transform-then-gather (used only to keep big
    oracle cases fast in tests; ``bench.py`` times :func:`rgcn_conv`)."""
    num_bases, cin, cout = basis.shape
    num_rel = att.shape[0]
    w = torch.matmul(att, basis.view(num_bases, -1)).view(num_rel, cin, cout)
    xw = torch.einsum('nf,rfo->nro', x, w)                           # [N, R, out]
    src, dst = edge_index[0], edge_index[1]
    msg = xw[src, edge_type]
    aggr = torch.zeros(x.shape[0], cout, dtype=x.dtype).index_add_(0, dst, msg)
    return aggr + torch.matmul(x, root) + bias


def dropout_adj(edge_index, edge_attr, p=0.5, force_undirected=False, num_nodes=None,
                training=True, mask=None, generator=None):
    """PyG-1.4.2 ``dropout_adj``: Bernoulli(1-p) keep mask per directed edge, no rescale.

    ``mask`` (bool, over the edges that are candidates for dropping -- all edges,
    or the ``row < col`` half when ``force_undirected``) injects a fixed mask for
    parity tests instead of drawing one.
    """
    if not training:
        return edge_index, edge_attr
    row, col = edge_index[0], edge_index[1]
    if force_undirected:
        sel = row < col
        row, col, edge_attr = row[sel], col[sel], edge_attr[sel]
    if mask is None:
        probs = torch.full((row.shape[0],), 1.0 - p, dtype=torch.float)
        mask = torch.bernoulli(probs, generator=generator).to(torch.bool)
    row, col, edge_attr = row[mask], col[mask], edge_attr[mask]
    if force_undirected:
        edge_index = torch.stack([torch.cat([row, col]), torch.cat([col, row])], 0)
        edge_attr = torch.cat([edge_attr, edge_attr])
        n = int(num_nodes)
        key = edge_index[0] * n + edge_index[1]          # coalesce == sort by (row, col)
        order = torch.argsort(key)
        edge_index, edge_attr = edge_index[:, order], edge_attr[order]
    else:
        edge_index = torch.stack([row, col], 0)
    return edge_index, edge_attr


class Batch(object):
    """``Batch.from_data_list``: keys containing 'index' get cumulative node offsets and are
    concatenated on the last dim; the rest on dim 0; ``batch`` = graph id per node."""

    @staticmethod
    def from_data_list(data_list):
        b = Batch()
        xs, eis, ets, ys, bvec, uf, vf = [], [], [], [], [], [], []
        off = 0
        for g, d in enumerate(data_list):
            n = d.x.shape[0]
            xs.append(d.x)
            eis.append(d.edge_index + off)
            ets.append(d.edge_type)
            ys.append(d.y)
            bvec.append(torch.full((n,), g, dtype=torch.long))
            if hasattr(d, 'u_feature'):
                uf.append(d.u_feature)
                vf.append(d.v_feature)
            off += n
        b.x = torch.cat(xs, 0)
        b.edge_index = torch.cat(eis, 1)
        b.edge_type = torch.cat(ets, 0)
        b.y = torch.cat(ys, 0)
        b.batch = torch.cat(bvec, 0)
        b.num_graphs = len(data_list)
        if uf:
            b.u_feature = torch.cat(uf, 0)
            b.v_feature = torch.cat(vf, 0)
        return b


# --------------------------------------------------------------------------- model
class RGCNConvRef(torch.nn.Module):
    """Parameter container with PyG-1.4.2 names/shapes/init (``basis, att, root, bias``)."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases = num_relations, num_bases
        self.basis = torch.nn.Parameter(torch.empty(num_bases, in_channels, out_channels))
        self.att = torch.nn.Parameter(torch.empty(num_relations, num_bases))
        self.root = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        size = self.num_bases * self.in_channels
        for p in (self.basis, self.att, self.root, self.bias):
            pyg_uniform_(size, p)

    def forward(self, x, edge_index, edge_type, fast=False):
        fn = rgcn_conv_fast if fast else rgcn_conv
        return fn(x, edge_index, edge_type, self.basis, self.att, self.root, self.bias)


class IGMCRef(torch.nn.Module):
    """``models.IGMC`` (ref ``models.py:170-217``) on top of the restated operators."""

    def __init__(self, num_features, latent_dim=(32, 32, 32, 32), num_relations=5, num_bases=4,
                 adj_dropout=0.2, force_undirected=False, side_features=False,
                 n_side_features=0, multiply_by=1, fast=False):
        super().__init__()
        self.adj_dropout = adj_dropout
        self.force_undirected = force_undirected
        self.multiply_by = multiply_by
        self.side_features = side_features
        self.fast = fast
        self.convs = torch.nn.ModuleList()
        dims = [num_features] + list(latent_dim)
        for i in range(len(latent_dim)):
            self.convs.append(RGCNConvRef(dims[i], dims[i + 1], num_relations, num_bases))
        width = 2 * sum(latent_dim) + (n_side_features if side_features else 0)
        self.lin1 = torch.nn.Linear(width, 128)
        self.lin2 = torch.nn.Linear(128, 1)

    def reset_parameters(self):
        for c in self.convs:
            c.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def forward(self, data, edge_mask=None, lin_mask=None):
        """``edge_mask`` / ``lin_mask`` inject the two dropout masks (parity tests)."""
        x, edge_index, edge_type = data.x, data.edge_index, data.edge_type
        if self.adj_dropout > 0:
            edge_index, edge_type = dropout_adj(
                edge_index, edge_type, p=self.adj_dropout, force_undirected=self.force_undirected,
                num_nodes=len(x), training=self.training, mask=edge_mask)
        states = []
        for conv in self.convs:
            x = torch.tanh(conv(x, edge_index, edge_type, fast=self.fast))
            states.append(x)
        states = torch.cat(states, 1)
        users = data.x[:, 0] == 1
        items = data.x[:, 1] == 1
        x = torch.cat([states[users], states[items]], 1)
        if self.side_features:
            x = torch.cat([x, data.u_feature.to(x.dtype), data.v_feature.to(x.dtype)], 1)
        x = F.relu(self.lin1(x))
        if self.training:
            if lin_mask is not None:
                x = x * lin_mask.to(x.dtype) * 2.0          # F.dropout(p=0.5): keep * 1/(1-p)
            else:
                x = F.dropout(x, p=0.5, training=True)
        x = self.lin2(x)
        return x[:, 0] * self.multiply_by


def global_sort_pool(x, batch, k):
    """PyG 1.4.2 ``torch_geometric.nn.glob.sort.global_sort_pool``: per graph, nodes sorted by the LAST channel
    (descending), the first ``k`` rows kept, graphs with fewer nodes padded with zero rows; returns ``[B, k * D]``.
    (The original pads with ``x.min() - 1``, sorts, and zeroes every element equal to the fill value afterwards --
    the same thing; ties are taken in node order here: ``stable=True``.)"""
    B = int(batch.max()) + 1
    D = x.shape[1]
    out = x.new_zeros(B, k, D)
    for g in range(B):
        rows = x[batch == g]
        order = torch.sort(rows[:, -1], descending=True, stable=True)[1][:k]
        out[g, :len(order)] = rows[order]
    return out.view(B, k * D)


class DGCNNRSRef(torch.nn.Module):
    """``models.DGCNN_RS`` (ref ``models.py:123-167`` on the ``DGCNN`` base ``:63-120``): R-GCN layers with
    latent_dim [32, 32, 32, 1], sort-pool readout, Conv1d / MaxPool1d / Conv1d, 2-layer MLP."""

    def __init__(self, num_features, latent_dim=(32, 32, 32, 1), k=30, num_relations=5, num_bases=4, adj_dropout=0.2,
                 force_undirected=False, fast=False):
        super().__init__()
        self.adj_dropout, self.force_undirected, self.fast = adj_dropout, force_undirected, fast
        self.k = int(k)
        self.convs = torch.nn.ModuleList()
        dims = [num_features] + list(latent_dim)
        for i in range(len(latent_dim)):
            self.convs.append(RGCNConvRef(dims[i], dims[i + 1], num_relations, num_bases))
        total = sum(latent_dim)
        self.conv1d_params1 = torch.nn.Conv1d(1, 16, total, total)
        self.maxpool1d = torch.nn.MaxPool1d(2, 2)
        self.conv1d_params2 = torch.nn.Conv1d(16, 32, 5, 1)
        dense_dim = int((self.k - 2) / 2 + 1)
        self.dense_dim = (dense_dim - 5 + 1) * 32
        self.lin1 = torch.nn.Linear(self.dense_dim, 128)
        self.lin2 = torch.nn.Linear(128, 1)

    def forward(self, data, edge_mask=None, lin_mask=None):
        x, edge_index, edge_type, batch = data.x, data.edge_index, data.edge_type, data.batch
        if self.adj_dropout > 0:
            edge_index, edge_type = dropout_adj(
                edge_index, edge_type, p=self.adj_dropout, force_undirected=self.force_undirected,
                num_nodes=len(x), training=self.training, mask=edge_mask)
        states = []
        for conv in self.convs:
            x = torch.tanh(conv(x, edge_index, edge_type, fast=self.fast))
            states.append(x)
        x = global_sort_pool(torch.cat(states, 1), batch, self.k).unsqueeze(1)
        x = F.relu(self.conv1d_params1(x))
        x = self.maxpool1d(x)
        x = F.relu(self.conv1d_params2(x))
        x = x.view(len(x), -1)
        x = F.relu(self.lin1(x))
        if self.training:
            if lin_mask is not None:
                x = x * lin_mask.to(x.dtype) * 2.0
            else:
                x = F.dropout(x, p=0.5, training=True)
        x = self.lin2(x)
        return x[:, 0]


def arr_loss(model):
    """Adjacent-rating regulariser, ref ``train_eval.py:167-174`` (without the ARR factor)."""
    total = 0
    for g in model.convs:
        w = torch.matmul(g.att, g.basis.view(g.num_bases, -1)).view(
            g.num_relations, g.in_channels, g.out_channels)
        total = total + torch.sum((w[1:] - w[:-1]) ** 2)
    return total


def loss_and_grads(model, batch, ARR=0.001, edge_mask=None, lin_mask=None):
    """One forward/backward of ref ``train_eval.py:158-175``; returns (loss, out, {name: grad})."""
    model.train()
    model.zero_grad()
    out = model(batch, edge_mask=edge_mask, lin_mask=lin_mask)
    loss = F.mse_loss(out, batch.y.view(-1).to(out.dtype))
    if ARR != 0:
        loss = loss + ARR * arr_loss(model)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    return loss.detach(), out.detach(), grads


def train_step(model, optimizer, batch, ARR=0.001, edge_mask=None, lin_mask=None):
    """ref ``train_eval.py:158-177`` for one batch."""
    model.train()
    optimizer.zero_grad()
    out = model(batch, edge_mask=edge_mask, lin_mask=lin_mask)
    loss = F.mse_loss(out, batch.y.view(-1).to(out.dtype))
    if ARR != 0:
        loss = loss + ARR * arr_loss(model)
    loss.backward()
    optimizer.step()
    return float(loss.detach())


def eval_sse(model, batch):
    """ref ``train_eval.py:182-199``: sum of squared errors of one batch, eval mode."""
    model.eval()
    with torch.no_grad():
        out = model(batch)
        return float(F.mse_loss(out, batch.y.view(-1).to(out.dtype), reduction='sum')), out
