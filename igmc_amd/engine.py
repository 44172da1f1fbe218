"""Pointer-level Python handles over the C ABI (``include/igmc_hip.h``).

Everything here works on raw device addresses (ints), so it is independent of how the buffers
were allocated (torch tensors in the product; the kernel-logic tests drive the host emulation
build of the same sources with numpy buffers).  The reference-shaped API lives one level up in
``igmc_amd.util_functions`` / ``igmc_amd.models`` / ``igmc_amd.train_eval``.
"""
import ctypes as C

import numpy as np

from . import _lib


def _p(x):
    """int / None / numpy array -> c_void_p"""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    return C.c_void_p(int(x))


class Graph(object):
    """Rating graph resident in HBM (replaces SparseRowIndexer/SparseColIndexer,
    reference util_functions.py:20-66)."""

    def __init__(self, A, device=0, lib=None):
        self.lib = lib or _lib.load()
        A = A.tocsr()
        A.sum_duplicates()
        A.sort_indices()
        self.n_users, self.n_items = A.shape
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        vals = np.asarray(A.data)
        if len(vals) and (vals.min() < 0 or vals.max() > 255 or np.any(vals != np.round(vals))):
            raise ValueError('adjacency values must be rating-label + 1 (small non-negative integers)')
        rating = np.ascontiguousarray(vals, dtype=np.uint8)
        self.nnz = int((rating != 0).sum())
        self.max_rel = int(rating.max()) - 1 if len(rating) else 0
        h = C.c_void_p()
        self.lib.call('igmc_graph_create', self.n_users, self.n_items, len(indices), _p(indptr), _p(indices),
                      _p(rating), device, C.byref(h))
        self.handle = h
        self.device = device

    def hbm_bytes(self):
        return int(self.lib.igmc_graph_hbm_bytes(self.handle))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.igmc_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch(object):
    """Arena for one extracted + collated batch of enclosing subgraphs."""

    def __init__(self, graph, max_graphs, hop=1, max_nodes_per_hop=None):
        self.lib = graph.lib
        self.graph = graph
        self.max_graphs = int(max_graphs)
        self.hop = int(hop)
        self.mnph = -1 if max_nodes_per_hop is None else int(max_nodes_per_hop)
        h = C.c_void_p()
        self.lib.call('igmc_batch_create', graph.handle, self.max_graphs, self.hop, self.mnph, C.byref(h))
        self.handle = h
        info = self.info()
        self.node_capacity, self.edge_capacity = info.node_capacity, info.edge_capacity
        self.num_labels = info.num_labels
        self.B = 0

    def extract(self, link_u, link_v, link_y, link_idx, first, B, sample_ratio=1.0, seed=0, epoch=0, stream=None):
        self.lib.call('igmc_extract_batch', self.graph.handle, self.handle, _p(link_u), _p(link_v), _p(link_y),
                      _p(link_idx), int(first), int(B), float(sample_ratio), int(seed) & (2 ** 64 - 1),
                      int(epoch) & (2 ** 64 - 1), _p(stream))
        self.B = int(B)

    def extract_cached(self, cache, link_y, link_idx, first, B, stream=None):
        """Batch ``link_idx[first:first+B]`` from a device-resident node-set cache (``igmc_extract_batch_cached``);
        ``cache`` = dict of device addresses uoff, unodes, udist, voff, vnodes, vdist."""
        self.lib.call('igmc_extract_batch_cached', self.graph.handle, self.handle, _p(cache['uoff']), _p(cache['unodes']),
                      _p(cache['udist']), _p(cache['voff']), _p(cache['vnodes']), _p(cache['vdist']), _p(link_y),
                      _p(link_idx), int(first), int(B), _p(stream))
        self.B = int(B)

    def extract_replay(self, u_lists, v_lists, u_dists, v_dists, ys, stream=None):
        """Parity mode: node sets given per graph (target first)."""
        B = len(u_lists)
        uoff = np.zeros(B + 1, np.int32)
        voff = np.zeros(B + 1, np.int32)
        uoff[1:] = np.cumsum([len(x) for x in u_lists])
        voff[1:] = np.cumsum([len(x) for x in v_lists])
        un = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in u_lists]))
        vn = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int32) for x in v_lists]))
        ud = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint8) for x in u_dists]))
        vd = np.ascontiguousarray(np.concatenate([np.asarray(x, np.uint8) for x in v_dists]))
        y = np.ascontiguousarray(np.asarray(ys, np.float32))
        self.lib.call('igmc_extract_batch_replay', self.graph.handle, self.handle, B, _p(un), _p(ud), _p(uoff),
                      _p(vn), _p(vd), _p(voff), _p(y), _p(stream))
        self.B = B

    def edge_dropout(self, p, force_undirected=False, seed=0, step=0, stream=None):
        self.lib.call('igmc_batch_edge_dropout', self.handle, float(p), int(bool(force_undirected)),
                      int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), _p(stream))

    def set_edge_flags(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self.lib.call('igmc_batch_set_edge_flags', self.handle, _p(flags), len(flags))

    def clear_edge_flags(self):
        self.lib.call('igmc_batch_clear_edge_flags', self.handle)

    def set_side_features(self, ptr, n_side):
        self.lib.call('igmc_batch_set_side_features', self.handle, _p(ptr), int(n_side))

    def bind_side_source(self, ptr, n_side):
        """Dataset-wide [n_links, n_side] side-feature matrix: every later ``extract`` gathers the batch's rows on the
        device (``igmc_batch_bind_side_source``)."""
        self.lib.call('igmc_batch_bind_side_source', self.handle, _p(ptr), int(n_side))

    def dense_layers(self, ws):
        """True when the dense per-layer kernels take the conv layers of this arena with workspace ``ws``."""
        return bool(self.lib.cdll.igmc_model_dense_layers(ws.handle, self.handle, int(self.B or self.max_graphs)))

    def want_transposed(self):
        """Keep the transposed dense blocks too (dense-layer kernels for models the subgraph kernel does not take)."""
        self.lib.call('igmc_batch_want_transposed', self.handle)

    def set_lean(self, lean=True):
        """Lean extraction: stop after the dense induced blocks (what the matrix-core subgraph kernel reads); the
        collated CSR is emitted on demand (``igmc_batch_set_lean``)."""
        self.lib.call('igmc_batch_set_lean', self.handle, int(bool(lean)))

    def assume_size(self, B):
        """The arena holds ``B`` subgraphs extracted by a replayed launch (``igmc_batch_assume_size``)."""
        self.lib.call('igmc_batch_assume_size', self.handle, int(B))
        self.B = int(B)

    def info(self, stream=None):
        info = _lib.BatchInfo()
        self.lib.call('igmc_batch_get_info', self.handle, C.byref(info), _p(stream))
        return info

    def device_ptr(self, name):
        return self.lib.igmc_batch_device_ptr(self.handle, _lib.BUF[name])

    def download(self, stream=None):
        """Host copy of the collated batch (synchronises)."""
        info = self.info(stream)
        if info.overflow:
            raise RuntimeError('batch arena overflow: N=%d E=%d exceed the capacity (%d, %d)' % (
                info.num_nodes, info.num_edges, info.node_capacity, info.edge_capacity))
        B, N, E = info.num_graphs, info.num_nodes, info.num_edges
        out = dict(
            node_off=np.zeros(B + 1, np.int32), n_users=np.zeros(B, np.int32),
            node_label=np.zeros(N, np.uint8), node_gid=np.zeros(N, np.int32), node_graph=np.zeros(N, np.int32),
            row_ptr=np.zeros(N + 1, np.int32), col=np.zeros(E, np.int32), erel=np.zeros(E, np.uint8),
            elab=np.zeros(E, np.uint8), eflag=np.zeros(E, np.uint8), y=np.zeros(B, np.float32))
        self.lib.call('igmc_batch_download', self.handle, _p(out['node_off']), _p(out['n_users']),
                      _p(out['node_label']), _p(out['node_gid']), _p(out['node_graph']), _p(out['row_ptr']),
                      _p(out['col']), _p(out['erel']), _p(out['elab']), _p(out['eflag']), _p(out['y']), _p(stream))
        out['B'], out['N'], out['E'] = B, N, E
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.igmc_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchSet(object):
    """A group of arenas of one geometry that ``igmc_extract_group`` fills in one launch per stage."""

    def __init__(self, arenas):
        self.arenas = list(arenas)
        self.lib = self.arenas[0].lib
        hs = (C.c_void_p * len(self.arenas))(*[a.handle for a in self.arenas])
        h = C.c_void_p()
        self.lib.call('igmc_batch_set_create', C.cast(hs, C.c_void_p), len(self.arenas), C.byref(h))
        self.handle = h

    def extract(self, count, link_u, link_v, link_y, link_idx, sel0, B, sample_ratio=1.0, seed=0, drop_p=0.0,
                force_undirected=False, drop_seed=0, stream=None):
        """Batches sel0 + 2 i (i < count; selectors of the device-side step control) into arenas 0 .. count-1, followed by
        their edge dropout when ``drop_p`` > 0."""
        self.lib.call('igmc_extract_group', self.arenas[0].graph.handle, self.handle, int(count), _p(link_u), _p(link_v),
                      _p(link_y), _p(link_idx), int(sel0), int(B), float(sample_ratio), int(seed) & (2 ** 64 - 1),
                      float(drop_p), int(bool(force_undirected)), int(drop_seed) & (2 ** 64 - 1), _p(stream))
        for a in self.arenas[:count]:
            a.B = int(B)

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.cdll.igmc_batch_set_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModelWorkspace(object):
    """Model geometry + activation/gradient workspace; knows the flat parameter layout."""

    PARAM_KINDS = ('BASIS', 'ROOT', 'BIAS', 'ATT')

    def __init__(self, lib, device, num_relations, num_bases, num_labels, n_side, max_nodes, max_edges, max_graphs):
        self.lib = lib
        self.R, self.Bs, self.L, self.S = int(num_relations), int(num_bases), int(num_labels), int(n_side)
        self.max_nodes, self.max_edges, self.max_graphs = int(max_nodes), int(max_edges), int(max_graphs)
        h = C.c_void_p()
        lib.call('igmc_model_create', int(device), self.R, self.Bs, self.L, self.S, self.max_nodes,
                 self.max_edges, self.max_graphs, C.byref(h))
        self.handle = h
        self.n_params = int(lib.igmc_param_count(h))

    def layout(self):
        """[(state_dict key, offset, shape)] in flat-buffer order (reference state_dict names)."""
        out = []
        for l in range(4):
            fin = self.L if l == 0 else 32
            shapes = dict(BASIS=(self.Bs, fin, 32), ROOT=(fin, 32), BIAS=(32,), ATT=(self.R, self.Bs))
            for kind, key in (('BASIS', 'basis'), ('ROOT', 'root'), ('BIAS', 'bias'), ('ATT', 'att')):
                cnt = C.c_int64()
                off = self.lib.igmc_param_offset(self.handle, l, _lib.P[kind], C.byref(cnt))
                assert cnt.value == int(np.prod(shapes[kind]))
                out.append(('convs.%d.%s' % (l, key), int(off), shapes[kind]))
        D = 256 + self.S
        for kind, key, shape in (('LIN1_W', 'lin1.weight', (128, D)), ('LIN1_B', 'lin1.bias', (128,)),
                                 ('LIN2_W', 'lin2.weight', (1, 128)), ('LIN2_B', 'lin2.bias', (1,))):
            cnt = C.c_int64()
            off = self.lib.igmc_param_offset(self.handle, 0, _lib.P[kind], C.byref(cnt))
            assert cnt.value == int(np.prod(shape))
            out.append((key, int(off), shape))
        return out

    def forward(self, params, batch, out, training=False, use_edge_flags=False, lin_mask=None, seed=0, step=0,
                multiply_by=1.0, stream=None):
        self.lib.call('igmc_model_forward', self.handle, _p(params), batch.handle, int(bool(training)),
                      int(bool(use_edge_flags)), _p(lin_mask), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1),
                      float(multiply_by), _p(out), _p(stream))

    def backward(self, params, batch, gout, grad, multiply_by=1.0, stream=None):
        self.lib.call('igmc_model_backward', self.handle, _p(params), batch.handle, _p(gout), float(multiply_by),
                      _p(grad), _p(stream))

    def loss_grad(self, params, batch, out, grad, loss, use_edge_flags=False, lin_mask=None, seed=0, step=0,
                  multiply_by=1.0, ARR=0.0, grad_scale=None, arr_scale=1.0, stream=None):
        if grad_scale is None:
            grad_scale = 1.0 / batch.B
        self.lib.call('igmc_model_loss_grad', self.handle, _p(params), batch.handle, int(bool(use_edge_flags)),
                      _p(lin_mask), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), float(multiply_by),
                      float(ARR), float(grad_scale), float(arr_scale), _p(out), _p(grad), _p(loss), _p(stream))

    def dense_path(self, batch, B):
        """True when forward / loss_grad / train_step on (batch arena, B) take the matrix-core subgraph kernel."""
        return bool(self.lib.igmc_model_dense_path(self.handle, batch.handle, int(B)))

    def step_form(self, batch, B):
        """Kernels of a training step on (batch arena, B): 1 subgraph kernel, 2 dense-layer kernels, 3 those in their
        group-split form, 0 per-layer kernels (igmc_model_step_form)."""
        return int(self.lib.igmc_model_step_form(self.handle, batch.handle, int(B)))

    def adam_step(self, params, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
                  weight_decay=0.0, stream=None):
        self.lib.call('igmc_adam_step', _p(params), _p(grad), _p(exp_avg), _p(exp_avg_sq), self.n_params, int(step),
                      float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), _p(stream))

    def sse_accumulate(self, out, batch, acc, stream=None, ctrl=None):
        """``ctrl``: the control block of a grouped evaluation pipeline -- its tick in the same launch."""
        if ctrl is not None:
            self.lib.call('igmc_sse_accumulate_tick', _p(out), batch.handle, _p(acc), _p(ctrl), _p(stream))
        else:
            self.lib.call('igmc_sse_accumulate', _p(out), batch.handle, _p(acc), _p(stream))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.igmc_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def adam_step(lib, params, grad, exp_avg, exp_avg_sq, n, step, lr, beta1=0.9, beta2=0.999, eps=1e-8,
              weight_decay=0.0, stream=None):
    """Fused Adam over a flat buffer (no model workspace needed)."""
    lib.call('igmc_adam_step', _p(params), _p(grad), _p(exp_avg), _p(exp_avg_sq), int(n), int(step), float(lr),
             float(beta1), float(beta2), float(eps), float(weight_decay), _p(stream))


def profile_enable(lib, on):
    lib.igmc_profile_enable(int(bool(on)))


def profile_fetch(lib, cap=64):
    names = ((C.c_char * 48) * cap)()
    ms = (C.c_float * cap)()
    calls = (C.c_int * cap)()
    n = lib.igmc_profile_fetch(C.cast(names, C.c_void_p), C.cast(ms, C.c_void_p), C.cast(calls, C.c_void_p), cap)
    return [(names[i].value.decode(), float(ms[i]), int(calls[i])) for i in range(max(n, 0))]


class SortPoolWorkspace(object):
    """Sort-pool readout (DGCNN_RS, reference ``models.py:63-167``) on top of a :class:`ModelWorkspace`'s conv kernels."""

    KEYS = ['convs.%d.%s' % (l, k) for l in range(4) for k in ('basis', 'root', 'bias', 'att')] + [
        'conv1d_params1.weight', 'conv1d_params1.bias', 'conv1d_params2.weight', 'conv1d_params2.bias',
        'lin1.weight', 'lin1.bias', 'lin2.weight', 'lin2.bias']

    def __init__(self, ws, k, max_nodes_per_graph):
        self.ws, self.lib = ws, ws.lib
        h = C.c_void_p()
        self.lib.call('igmc_sortpool_create', ws.handle, int(k), int(max_nodes_per_graph), C.byref(h))
        self.handle = h
        lay = (C.c_int64 * 27)()
        self.lib.call('igmc_sortpool_layout', h, C.cast(lay, C.c_void_p))
        self.offsets = dict(zip(self.KEYS, [int(x) for x in lay[:24]]))
        self.n_params, self.dense, self.k = int(lay[24]), int(lay[25]), int(lay[26])

    def __del__(self):
        try:
            self.lib.cdll.igmc_sortpool_destroy(self.handle)
        except Exception:
            pass

    def forward(self, params, batch, out, training=False, use_edge_flags=False, lin_mask=None, seed=0, step=0, stream=None):
        self.lib.call('igmc_sortpool_forward', self.handle, _p(params), batch.handle, int(bool(training)),
                      int(bool(use_edge_flags)), _p(lin_mask), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1),
                      _p(out), _p(stream))

    def loss_grad(self, params, batch, out, grad, loss, use_edge_flags=False, lin_mask=None, seed=0, step=0, ARR=0.0,
                  grad_scale=None, arr_scale=1.0, stream=None):
        if grad_scale is None:
            grad_scale = 1.0 / batch.B
        self.lib.call('igmc_sortpool_loss_grad', self.handle, _p(params), batch.handle, int(bool(use_edge_flags)),
                      _p(lin_mask), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), float(ARR), float(grad_scale),
                      float(arr_scale), _p(out), _p(grad), _p(loss), _p(stream))
