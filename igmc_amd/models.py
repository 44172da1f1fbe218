"""Drop-in mirror of the reference's ``models.IGMC`` (reference ``models.py:170-217``) on the gfx950 engine.

Same constructor arguments, same attributes used by the reference's train loop
(``.convs[i].att / .basis / .num_bases / .num_relations / .in_channels / .out_channels``,
``.lin1``, ``.lin2``, ``.reset_parameters()``, ``.multiply_by``) and the same ``state_dict`` keys and shapes
(``convs.{l}.{basis,att,root,bias}``, ``lin1.{weight,bias}``, ``lin2.{weight,bias}``), so checkpoints are
interchangeable with the reference (``Main.py:36-45``, ``--transfer``, ``--continue-from``).

All parameters are views into ONE flat fp32 buffer (the layout the HIP kernels, the fused Adam and the
RCCL gradient all-reduce work on).  ``forward`` runs the hand-written HIP kernels through the C ABI; it is
differentiable (custom autograd function) so foreign training loops / torch optimisers also work, while
``igmc_amd.train_eval`` uses the fused loss+gradient+Adam path.  No PyTorch-Geometric, no CPU fallback.
"""
import math

import torch
import torch.nn as nn

from . import _lib, engine
from .util_functions import DeviceBatch


class RGCNConv(nn.Module):
    """Parameter container with PyG-1.4.2 ``RGCNConv`` names / shapes / init (the message passing itself
    happens inside the fused kernels).  Registration order basis, att, root, bias = PyG's."""

    def __init__(self, in_channels, out_channels, num_relations, num_bases):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_relations, self.num_bases = num_relations, num_bases
        self.basis = nn.Parameter(torch.empty(num_bases, in_channels, out_channels))
        self.att = nn.Parameter(torch.empty(num_relations, num_bases))
        self.root = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        # torch_geometric.nn.inits.uniform(size, tensor): U(-1/sqrt(size), 1/sqrt(size)), size = num_bases*in
        bound = 1.0 / math.sqrt(self.num_bases * self.in_channels)
        with torch.no_grad():
            for p in (self.basis, self.att, self.root, self.bias):
                p.uniform_(-bound, bound)

    def __repr__(self):
        return '{}({}, {}, num_relations={})'.format(self.__class__.__name__, self.in_channels, self.out_channels,
                                                     self.num_relations)


class _IGMCFunction(torch.autograd.Function):
    """out = IGMC(batch; params).  Inputs: the model + batch (non-tensor) and every parameter view."""

    @staticmethod
    def forward(ctx, model, data, training, *params):
        ws = model._workspace(data)
        st = torch.cuda.current_stream().cuda_stream
        out = torch.empty(data.num_graphs, dtype=torch.float32, device=model._flat.device)
        use_flags = bool(training and model.adj_dropout > 0)
        model._step += 1
        if use_flags:
            data.arena.edge_dropout(model.adj_dropout, model.force_undirected, model.seed, model._step, st)
        ws.forward(model._flat.data_ptr(), data.arena, out.data_ptr(), training=training, use_edge_flags=use_flags,
                   seed=model.seed, step=model._step, multiply_by=float(model.multiply_by), stream=st)
        ctx.model, ctx.data = model, data
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, gout):
        model, data = ctx.model, ctx.data
        if not ctx.training:
            raise RuntimeError('IGMC backward needs model.train() (eval-mode forward keeps no activations)')
        ws = model._workspace(data)
        st = torch.cuda.current_stream().cuda_stream
        grad = torch.empty_like(model._flat)
        gout = gout.contiguous().float()
        ws.backward(model._flat.data_ptr(), data.arena, gout.data_ptr(), grad.data_ptr(),
                    multiply_by=float(model.multiply_by), stream=st)
        where = {k: (o, n, shape) for (k, o, n, shape) in model._views}
        grads = []
        for key, _ in model.named_parameters():       # same order as the *params inputs
            o, n, shape = where[key]
            grads.append(grad[o:o + n].view(shape))
        return (None, None, None) + tuple(grads)


class IGMC(nn.Module):
    """The GNN model of Inductive Graph-based Matrix Completion: 4x R-GCN (+tanh) + centre-node readout +
    2-layer MLP regressor (reference ``models.py:170-217``)."""

    def __init__(self, dataset, gconv=RGCNConv, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=2,
                 regression=False, adj_dropout=0.2, force_undirected=False, side_features=False,
                 n_side_features=0, multiply_by=1, seed=0):
        super().__init__()
        if list(latent_dim) != [32, 32, 32, 32]:
            raise NotImplementedError('the gfx950 kernels are built for latent_dim=[32,32,32,32] (reference Main.py:391)')
        if num_bases != 4:
            raise NotImplementedError('the gfx950 kernels are built for num_bases=4 (reference Main.py:393)')
        if not regression:
            raise NotImplementedError('only regression=True (what reference Main.py:392 uses) is accelerated')
        if gconv is not RGCNConv and getattr(gconv, '__name__', '') != 'RGCNConv':
            raise NotImplementedError('IGMC uses RGCNConv')
        _lib.load()      # fail loudly right here if the gfx950 library is missing
        self.regression = regression
        self.adj_dropout = adj_dropout
        self.force_undirected = force_undirected
        self.multiply_by = multiply_by
        self.side_features = side_features
        self.n_side_features = int(n_side_features) if side_features else 0
        self.num_relations, self.num_bases = int(num_relations), int(num_bases)
        self.num_features = int(dataset.num_features)
        self.seed = int(seed)
        self._step = 0
        self.convs = nn.ModuleList()
        self.convs.append(RGCNConv(self.num_features, latent_dim[0], num_relations, num_bases))
        for i in range(0, len(latent_dim) - 1):
            self.convs.append(RGCNConv(latent_dim[i], latent_dim[i + 1], num_relations, num_bases))
        self.lin1 = nn.Linear(2 * sum(latent_dim) + self.n_side_features, 128)
        self.lin2 = nn.Linear(128, 1)
        self._ws = {}
        self._flat = None
        self._flat_grad = None
        self._flatten()

    # ------------------------------------------------------------------ flat parameter buffer
    def _layout(self):
        """[(state_dict key, offset, numel, shape)] in the order of the C-ABI layout (igmc_hip.h)."""
        out, off = [], 0
        for l, conv in enumerate(self.convs):
            for key in ('basis', 'root', 'bias', 'att'):
                p = getattr(conv, key)
                out.append(('convs.%d.%s' % (l, key), off, p.numel(), tuple(p.shape)))
                off += p.numel()
        for mod, name in ((self.lin1, 'lin1'), (self.lin2, 'lin2')):
            for key in ('weight', 'bias'):
                p = getattr(mod, key)
                out.append(('%s.%s' % (name, key), off, p.numel(), tuple(p.shape)))
                off += p.numel()
        return out, off

    def _flatten(self, device=None):
        """(Re)build the flat buffer and make every Parameter a view of it."""
        views, total = self._layout()
        ref = self.lin2.bias
        device = device or ref.device
        flat = torch.empty(total, dtype=torch.float32, device=device)
        named = dict(self.named_parameters())
        with torch.no_grad():
            for key, off, n, shape in views:
                flat[off:off + n].copy_(named[key].detach().reshape(-1).to(device=device, dtype=torch.float32))
        for key, off, n, shape in views:
            named[key].data = flat[off:off + n].view(shape)
            named[key].grad = None
        self._flat, self._views = flat, views
        self._flat_grad = None

    def _apply(self, fn, *args, **kwargs):
        # .to(device) / .cuda() / .float(): move the flat buffer and re-create the views
        probe = fn(self._flat)
        out = super()._apply(fn, *args, **kwargs)
        self._flatten(device=probe.device)
        self._ws = {}
        return out

    def load_state_dict(self, state_dict, *args, **kwargs):
        """``nn.Module.load_state_dict`` + the engine's invariant that its exchange regions hold finite values only: steps
        that ran on diverged (non-finite) parameters may have left NaN rows in a subgraph slot, which every later gather of that
        slot would multiply by its zero block entries -- cleared whenever parameters are restored (``igmc_model_reset_exchange``)."""
        out = super().load_state_dict(state_dict, *args, **kwargs)
        if self._flat.device.type == 'cuda':
            st = torch.cuda.current_stream(self._flat.device).cuda_stream
            for ws in self._ws.values():
                ws.lib.call('igmc_model_reset_exchange', ws.handle, engine._p(st))
        return out

    def flat_parameters(self):
        return self._flat

    def flat_grad(self):
        """Flat gradient buffer; every ``param.grad`` is a view of it."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            named = dict(self.named_parameters())
            for key, off, n, shape in self._views:
                named[key].grad = self._flat_grad[off:off + n].view(shape)
        return self._flat_grad

    def reset_parameters(self):
        """reference ``models.py:31-35`` (called by ``train_multiple_epochs``, ``train_eval.py:53``)."""
        for conv in self.convs:
            conv.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    # ------------------------------------------------------------------ engine workspace
    def _workspace(self, data):
        if self._flat.device.type != 'cuda':
            raise RuntimeError('igmc_amd.IGMC runs on an MI355X only: call model.to("cuda") (there is no CPU path)')
        arena = data.arena
        key = (arena.node_capacity, arena.edge_capacity, arena.max_graphs, arena.num_labels)
        ws = self._ws.get(key)
        if ws is None:
            if arena.num_labels != self.num_features:
                raise ValueError('dataset hop (num_features=%d) does not match the model (%d)' % (
                    arena.num_labels, self.num_features))
            ws = engine.ModelWorkspace(_lib.load(), self._flat.device.index or 0, self.num_relations, self.num_bases,
                                       self.num_features, self.n_side_features, arena.node_capacity,
                                       arena.edge_capacity, arena.max_graphs)
            mine = [(k, o, s) for (k, o, n, s) in self._views]
            if sorted(ws.layout()) != sorted(mine):
                raise AssertionError('flat parameter layout differs between Python and the C ABI')
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ forward
    def forward(self, data):
        """``data``: a :class:`igmc_amd.util_functions.DeviceBatch` (what ``train_eval``'s loader yields)."""
        if not isinstance(data, DeviceBatch):
            raise TypeError('igmc_amd.IGMC consumes device-resident batches (igmc_amd.train_eval.DataLoader); '
                            'got %s' % type(data).__name__)
        if self.side_features and data.side is None:
            raise ValueError('model built with side_features=True but the dataset has no u/v features')
        params = [p for p in self.parameters()]
        if torch.is_grad_enabled() and self.training:
            return _IGMCFunction.apply(self, data, True, *params)
        with torch.no_grad():
            return _IGMCFunction.forward(_NoCtx(), self, data, bool(self.training), *params)

    def __repr__(self):
        return self.__class__.__name__


class _NoCtx(object):
    pass


class DGCNN_RS(IGMC):
    """DGCNN with R-GCN convolutions (reference ``models.py:123-167`` on the ``DGCNN`` base ``:63-120``; the variant the
    reference keeps behind ``if False`` at ``Main.py:364-380``): four R-GCN layers with ``latent_dim = [32, 32, 32, 1]``,
    ``global_sort_pool(k)`` over the 97 concatenated channels, ``Conv1d(1, 16, 97, 97)`` / ``MaxPool1d(2, 2)`` /
    ``Conv1d(16, 32, 5, 1)``, ``lin1`` / dropout / ``lin2``.

    Same constructor arguments, attributes and ``state_dict`` keys / shapes as the reference
    (``convs.{l}.{basis,att,root,bias}``, ``conv1d_params{1,2}.{weight,bias}``, ``lin{1,2}.{weight,bias}``).  The conv
    layers are the per-layer HIP kernels of IGMC (the 32 -> 1 layer as a zero-padded 32 -> 32 one), the readout and its
    backward are ``igmc_amd/csrc/sortpool.hip``.  Training goes through ``train_eval`` with ``FlatAdam``
    (``fused_loss_grad``); the differentiable-forward route of ``IGMC`` is not provided for this family."""

    def __init__(self, dataset, gconv=RGCNConv, latent_dim=[32, 32, 32, 1], k=30, num_relations=5, num_bases=2,
                 regression=False, adj_dropout=0.2, force_undirected=False, seed=0):
        nn.Module.__init__(self)
        if list(latent_dim) != [32, 32, 32, 1]:
            raise NotImplementedError('the gfx950 kernels are built for latent_dim=[32,32,32,1] (reference Main.py:368)')
        if num_bases != 4:
            raise NotImplementedError('the gfx950 kernels are built for num_bases=4 (reference Main.py:371)')
        if not regression:
            raise NotImplementedError('only regression=True (what reference Main.py:372 uses) is accelerated')
        if gconv is not RGCNConv and getattr(gconv, '__name__', '') != 'RGCNConv':
            raise NotImplementedError('DGCNN_RS uses RGCNConv')
        _lib.load()
        self.regression = regression
        self.adj_dropout, self.force_undirected = adj_dropout, force_undirected
        self.multiply_by, self.side_features, self.n_side_features = 1, False, 0
        self.num_relations, self.num_bases = int(num_relations), int(num_bases)
        self.num_features = int(dataset.num_features)
        self.seed = int(seed)
        self._step = 0
        if k < 1:       # percentile of the subgraph sizes -> number of pooled nodes (reference models.py:70-73)
            node_nums = sorted(self._subgraph_sizes(dataset))
            k = node_nums[int(math.ceil(k * len(node_nums))) - 1]
            k = max(10, k)
        self.k = int(k)
        print('k used in sortpooling is:', self.k)
        self.convs = nn.ModuleList()
        self.convs.append(RGCNConv(self.num_features, latent_dim[0], num_relations, num_bases))
        for i in range(0, len(latent_dim) - 1):
            self.convs.append(RGCNConv(latent_dim[i], latent_dim[i + 1], num_relations, num_bases))
        self.total_latent_dim = sum(latent_dim)
        self.conv1d_params1 = nn.Conv1d(1, 16, self.total_latent_dim, self.total_latent_dim)
        self.maxpool1d = nn.MaxPool1d(2, 2)
        self.conv1d_params2 = nn.Conv1d(16, 32, 5, 1)
        dense_dim = int((self.k - 2) / 2 + 1)
        self.dense_dim = (dense_dim - 5 + 1) * 32
        if self.dense_dim < 32:
            raise ValueError('k = %d leaves nothing after Conv1d(16, 32, 5)' % self.k)
        self.lin1 = nn.Linear(self.dense_dim, 128)
        self.lin2 = nn.Linear(128, 1)
        self._ws, self._sp = {}, {}
        self._flat = None
        self._flat_grad = None
        self._flatten()

    @staticmethod
    def _subgraph_sizes(dataset, max_samples=1000):
        """``[g.num_nodes for g in dataset]`` of the reference, on an evenly spaced sample of at most ``max_samples``
        links (the reference walks the whole dataset: minutes of extraction for a number that is a percentile)."""
        cache = getattr(dataset, '_cache_t', None)
        if cache is not None:       # static dataset: the node-set cache holds every subgraph's size -- exact, like the reference
            uo, vo = cache['uoff'].cpu().numpy(), cache['voff'].cpu().numpy()
            return [int(x) for x in (uo[1:] - uo[:-1]) + (vo[1:] - vo[:-1])]
        n = len(dataset)
        step = max(1, n // max_samples)
        return [int(dataset[i].num_nodes) for i in range(0, n, step)]

    def _layout(self):
        out, off = [], 0
        for l, conv in enumerate(self.convs):
            for key in ('basis', 'root', 'bias', 'att'):
                p = getattr(conv, key)
                out.append(('convs.%d.%s' % (l, key), off, p.numel(), tuple(p.shape)))
                off += p.numel()
        for mod, name in ((self.conv1d_params1, 'conv1d_params1'), (self.conv1d_params2, 'conv1d_params2'),
                          (self.lin1, 'lin1'), (self.lin2, 'lin2')):
            for key in ('weight', 'bias'):
                p = getattr(mod, key)
                out.append(('%s.%s' % (name, key), off, p.numel(), tuple(p.shape)))
                off += p.numel()
        return out, off

    def reset_parameters(self):
        """reference ``models.py:87-93``"""
        for conv in self.convs:
            conv.reset_parameters()
        self.conv1d_params1.reset_parameters()
        self.conv1d_params2.reset_parameters()
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._sp = {}
        return out

    def _sortpool(self, data):
        """(conv workspace, sort-pool workspace) for this batch arena."""
        if self._flat.device.type != 'cuda':
            raise RuntimeError('igmc_amd.DGCNN_RS runs on an MI355X only: call model.to("cuda") (there is no CPU path)')
        arena = data.arena
        key = (arena.node_capacity, arena.edge_capacity, arena.max_graphs, arena.num_labels)
        sp = self._sp.get(key)
        if sp is None:
            if arena.num_labels != self.num_features:
                raise ValueError('dataset hop (num_features=%d) does not match the model (%d)' % (
                    arena.num_labels, self.num_features))
            ws = engine.ModelWorkspace(_lib.load(), self._flat.device.index or 0, self.num_relations, self.num_bases,
                                       self.num_features, 0, arena.node_capacity, arena.edge_capacity, arena.max_graphs)
            sp = engine.SortPoolWorkspace(ws, self.k, max(2, arena.node_capacity // arena.max_graphs))
            mine = {k: o for (k, o, n, s) in self._views}
            if mine != sp.offsets or sp.n_params != self._flat.numel() or sp.dense != self.dense_dim:
                raise AssertionError('flat parameter layout differs between Python and the C ABI')
            self._ws[key], self._sp[key] = ws, sp
        return sp

    def _workspace(self, data):
        return self._sortpool(data).ws

    def forward_into(self, data, out, training=False, stream=None):
        """``out[:B]`` <- model(batch) (what ``train_eval.eval_loss`` calls; no host synchronisation)."""
        sp = self._sortpool(data)
        st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        use_flags = bool(training and self.adj_dropout > 0)
        if training:
            self._step += 1
            if use_flags:
                data.arena.edge_dropout(self.adj_dropout, self.force_undirected, self.seed, self._step, st)
        sp.forward(self._flat.data_ptr(), data.arena, out.data_ptr(), training=training, use_edge_flags=use_flags,
                   seed=self.seed, step=self._step, stream=st)

    def forward(self, data):
        if not isinstance(data, DeviceBatch):
            raise TypeError('igmc_amd.DGCNN_RS consumes device-resident batches (igmc_amd.train_eval.DataLoader); '
                            'got %s' % type(data).__name__)
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('DGCNN_RS trains through train_eval with FlatAdam (fused_loss_grad); wrap '
                                      'inference in torch.no_grad()')
        out = torch.empty(data.num_graphs, dtype=torch.float32, device=self._flat.device)
        self.forward_into(data, out, training=bool(self.training))
        return out

    def fused_loss_grad(self, data, ARR=0.0, grad_scale=None, arr_scale=1.0, lin_mask=None, loss=None, out=None):
        """One step's loss and gradient (reference ``train_eval.py:158-175``): MSE + ARR over ``self.convs``; the
        gradient lands in ``flat_grad()``.  Returns the 2-float device tensor [loss, sum of squared errors]."""
        sp = self._sortpool(data)
        st = torch.cuda.current_stream().cuda_stream
        dev = self._flat.device
        loss = loss if loss is not None else torch.zeros(2, dtype=torch.float32, device=dev)
        out = out if out is not None else torch.empty(data.num_graphs, dtype=torch.float32, device=dev)
        use_flags = self.adj_dropout > 0
        self._step += 1
        if use_flags:
            data.arena.edge_dropout(self.adj_dropout, self.force_undirected, self.seed, self._step, st)
        sp.loss_grad(self._flat.data_ptr(), data.arena, out.data_ptr(), self.flat_grad().data_ptr(), loss.data_ptr(),
                     use_edge_flags=use_flags, lin_mask=lin_mask, seed=self.seed, step=self._step, ARR=float(ARR),
                     grad_scale=grad_scale, arr_scale=arr_scale, stream=st)
        return loss


class GNN(IGMC):
    """Placeholder for the reference's GCN baseline (``models.py:12-60``; dead code behind ``if False`` at
    ``Main.py:364``).  Not part of the accelerated path."""

    def __init__(self, *a, **k):
        raise NotImplementedError('GNN / DGCNN (GCNConv message passing) are dead, broken code in the reference '
                                  '(Main.py:364; edge_type used before assignment, models.py:40,98); the R-GCN variant '
                                  'is igmc_amd.models.DGCNN_RS')


DGCNN = GNN
