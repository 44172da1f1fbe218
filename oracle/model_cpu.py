"""TEST INFRASTRUCTURE: ctypes wrapper of ``oracle/model_cpu.c``, the OpenMP host twin of the IGMC model step (SURVEY.md 8(b):
``igmc_cpu_*``).  Only tests, ``__graft_entry__`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

The twin takes a collated batch (``pyg_ref.Batch``: one-hot ``x``, ``edge_index``, ``edge_type``, ``batch``, ``y``) or the raw
output of the extraction twin, and the flat parameter buffer in the C ABI's layout (``flat_from_model`` builds it from an
``IGMCRef`` / any module with the reference's state-dict keys)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'model_cpu.c')
LIB = os.path.join(HERE, '_build', 'libmodel_cpu.so')


def build(force=False):
    """gcc -O3 -fopenmp -shared: the checker is BUILT by ``__graft_entry__.build()``; building it is not using it."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ['gcc', '-O3', '-mavx2', '-mfma', '-fopenmp', '-shared', '-fPIC', '-o', LIB, SRC, '-lm']     # (x86-64-v3: both boxes)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('build of the model twin failed: %s\n%s' % (' '.join(cmd), r.stdout.decode()))
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.igmc_cpu_param_count.restype = C.c_int64
    return _lib


def set_threads(n=0):
    return int(lib().igmc_cpu_model_set_threads(C.c_int(int(n))))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Config(object):
    """``dims`` = [input features, latent_dim ...] (reference models.py:176-184), ``R`` relations, ``NB`` bases, ``hid`` = 128."""

    def __init__(self, dims, R, NB=4, hid=128):
        self.dims = np.ascontiguousarray(dims, np.int32)
        self.nl, self.R, self.NB, self.hid = len(dims) - 1, int(R), int(NB), int(hid)
        self.n_params = int(lib().igmc_cpu_param_count(C.c_int(self.nl), _p(self.dims), C.c_int(self.R), C.c_int(self.NB),
                                                       C.c_int(self.hid)))
        assert self.n_params > 0

    def layout(self):
        """[(state-dict key, offset, shape)] -- the C ABI's order (include/igmc_hip.h: igmc_param_offset)."""
        out, off = [], 0
        width = 2 * int(self.dims[1:].sum())
        for l in range(self.nl):
            di, do = int(self.dims[l]), int(self.dims[l + 1])
            for key, shape in (('basis', (self.NB, di, do)), ('root', (di, do)), ('bias', (do,)), ('att', (self.R, self.NB))):
                out.append(('convs.%d.%s' % (l, key), off, shape))
                off += int(np.prod(shape))
        for key, shape in (('lin1.weight', (self.hid, width)), ('lin1.bias', (self.hid,)), ('lin2.weight', (1, self.hid)),
                           ('lin2.bias', (1,))):
            out.append((key, off, shape))
            off += int(np.prod(shape))
        assert off == self.n_params
        return out


def config_of(model):
    c0 = model.convs[0]
    dims = [int(c0.basis.shape[1])] + [int(c.basis.shape[2]) for c in model.convs]
    return Config(dims, int(c0.att.shape[0]), int(c0.att.shape[1]), int(model.lin1.weight.shape[0]))


def flat_from_model(model, cfg=None):
    cfg = cfg or config_of(model)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    flat = np.zeros(cfg.n_params, np.float32)
    for key, off, shape in cfg.layout():
        flat[off:off + int(np.prod(shape))] = sd[key].astype(np.float32).reshape(-1)
    return flat


def unflatten(cfg, flat):
    return {key: flat[off:off + int(np.prod(shape))].reshape(shape) for key, off, shape in cfg.layout()}


def collate_pyg(batch, edge_keep=None):
    """A ``pyg_ref.Batch`` as the twin's arrays.  ``edge_keep``: bool per directed edge (``dropout_adj`` without
    ``force_undirected``: the oracle's ``mask``)."""
    x = batch.x.numpy()
    assert np.all((x == 0) | (x == 1)) and np.all(x.sum(1) == 1), 'x is not one-hot'
    label = np.ascontiguousarray(x.argmax(1), np.int32)
    ei = batch.edge_index.numpy()
    et = batch.edge_type.numpy()
    if edge_keep is not None:
        keep = np.asarray(edge_keep, bool)
        ei, et = ei[:, keep], et[keep]
    bv = batch.batch.numpy()
    B = int(batch.num_graphs)
    ge = bv[ei[0]]
    order = np.argsort(ge, kind='stable')
    ei, et, ge = ei[:, order], et[order], ge[order]
    node_off = np.zeros(B + 1, np.int64)
    node_off[1:] = np.cumsum(np.bincount(bv, minlength=B))
    edge_off = np.zeros(B + 1, np.int64)
    edge_off[1:] = np.cumsum(np.bincount(ge, minlength=B))
    return dict(B=B, node_off=node_off, edge_off=edge_off, label=label, src=np.ascontiguousarray(ei[0], np.int32),
                dst=np.ascontiguousarray(ei[1], np.int32), rel=np.ascontiguousarray(et, np.uint8))


def collate_raw(raw, cap_u, cap_v):
    """The extraction twin's raw output (``extract_cpu.extract_batch(..., raw=True)``) as the twin's arrays, in C."""
    n_u, n_v, users, items, ulab, vlab, n_e, edges = raw
    B = len(n_u)
    N, E2 = int(n_u.sum() + n_v.sum()), int(2 * n_e.sum())
    node_off, edge_off = np.zeros(B + 1, np.int64), np.zeros(B + 1, np.int64)
    label, src, dst, rel = np.zeros(N, np.int32), np.zeros(E2, np.int32), np.zeros(E2, np.int32), np.zeros(E2, np.uint8)
    rc = lib().igmc_cpu_collate(C.c_int(B), C.c_int(cap_u), C.c_int(cap_v), C.c_int64(edges.shape[1]), _p(n_u), _p(n_v), _p(ulab),
                                _p(vlab), _p(n_e), _p(edges), _p(node_off), _p(edge_off), _p(label), _p(src), _p(dst), _p(rel))
    assert rc == 0, 'a link of the batch was not extracted (capacity)'
    return dict(B=B, node_off=node_off, edge_off=edge_off, label=label, src=src, dst=dst, rel=rel)


def loss_grad(cfg, flat, cb, y=None, lin_mask=None, multiply_by=1.0, ARR=0.001, want_grad=True):
    """-> (out [B], grad [n_params] or None, (loss, sse) or None).  ``lin_mask`` [B, hid] keep flags: training mode with that
    mask; None: eval mode (no dropout)."""
    B = cb['B']
    out = np.zeros(B, np.float32)
    grad = np.zeros(cfg.n_params, np.float32) if want_grad else None
    loss = np.zeros(2, np.float64) if y is not None else None
    yy = None if y is None else np.ascontiguousarray(y, np.float32)
    lm = None if lin_mask is None else np.ascontiguousarray(lin_mask, np.uint8).reshape(-1)
    flat = np.ascontiguousarray(flat, np.float32)
    rc = lib().igmc_cpu_model_loss_grad(C.c_int(cfg.nl), _p(cfg.dims), C.c_int(cfg.R), C.c_int(cfg.NB), C.c_int(cfg.hid), _p(flat),
                                        C.c_int(B), _p(cb['node_off']), _p(cb['label']), _p(cb['edge_off']), _p(cb['src']),
                                        _p(cb['dst']), _p(cb['rel']), _p(yy), _p(lm), C.c_int(1 if lm is not None else 0),
                                        C.c_float(multiply_by), C.c_float(ARR), _p(out), _p(grad), _p(loss))
    if rc != 0:
        raise RuntimeError('igmc_cpu_model_loss_grad: %d' % rc)
    return out, grad, (None if loss is None else (float(loss[0]), float(loss[1])))


def adam_step(flat, grad, exp_avg, exp_avg_sq, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In place on the four float32 arrays; ``t`` is 1-based."""
    for a in (flat, grad, exp_avg, exp_avg_sq):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    rc = lib().igmc_cpu_adam_step(_p(flat), _p(grad), _p(exp_avg), _p(exp_avg_sq), C.c_int64(flat.size), C.c_int64(t), C.c_float(lr),
                                  C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(weight_decay))
    assert rc == 0
