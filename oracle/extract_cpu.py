"""TEST INFRASTRUCTURE: ctypes wrapper of ``oracle/extract_cpu.c``, the OpenMP host twin of the engine's free-running subgraph
extraction (``igmc_cpu_*``).  Only tests, ``__graft_entry__`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

``extract_batch(A, links, first, B, ...)`` returns, per link, what ``helpers.graph_canonical`` returns for a graph of a
downloaded engine batch: ``(users, items, {gid: label}, {gid: label}, sorted (u_gid, v_gid, relation) triples)``."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'extract_cpu.c')
LIB = os.path.join(HERE, '_build', 'libextract_cpu.so')


def build(force=False):
    """gcc -O2 -fopenmp -shared: the checker is BUILT by ``__graft_entry__.build()``; building it is not using it."""
    deps = [SRC, os.path.join(ROOT, 'include', 'igmc_rng.h')]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'include'), '-o', LIB, SRC]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('build of the extraction twin failed: %s\n%s' % (' '.join(cmd), r.stdout.decode()))
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.igmc_cpu_extract_batch.restype = C.c_int
    return _lib


def set_threads(n=0):
    """OpenMP threads of the following calls (0: leave the default); returns the count in force."""
    return int(lib().igmc_cpu_set_threads(C.c_int(int(n))))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def prepare(A):
    """The CSR / CSC arrays of the rating matrix (done once per graph; ``extract_batch`` takes the result in place of ``A``)."""
    A = A.tocsr()
    A.sort_indices()
    At = A.tocsc()
    At.sort_indices()
    nu, nv = A.shape
    u_ptr, u_idx = A.indptr.astype(np.int64), A.indices.astype(np.int32)
    u_rel = (np.asarray(A.data).astype(np.int64) - 1).astype(np.uint8)
    v_ptr, v_idx = At.indptr.astype(np.int64), At.indices.astype(np.int32)
    return nu, nv, u_ptr, u_idx, u_rel, v_ptr, v_idx, int(A.nnz)


def extract_batch(A, link_u, link_v, first, B, hop=1, sample_ratio=1.0, max_nodes_per_hop=None, seed=0, epoch=0, link_idx=None,
                  cap_u=None, cap_v=None, raw=False):
    """``A``: scipy CSR (users x items) whose values are relation + 1 (the engine's graph, ``igmc_graph_create``)."""
    nu, nv, u_ptr, u_idx, u_rel, v_ptr, v_idx, nnz = A if isinstance(A, tuple) else prepare(A)
    lu, lv = np.ascontiguousarray(link_u, np.int32), np.ascontiguousarray(link_v, np.int32)
    li = None if link_idx is None else np.ascontiguousarray(link_idx, np.int64)
    mn = -1 if max_nodes_per_hop is None else int(max_nodes_per_hop)
    cap_u = int(cap_u if cap_u is not None else nu)
    cap_v = int(cap_v if cap_v is not None else nv)
    epl = int(min(nnz, cap_u * cap_v))          # (an induced subgraph holds at most one rating per node pair)
    n_u, n_v = np.zeros(B, np.int32), np.zeros(B, np.int32)
    users, items = np.zeros((B, cap_u), np.int32), np.zeros((B, cap_v), np.int32)
    ulab, vlab = np.zeros((B, cap_u), np.uint8), np.zeros((B, cap_v), np.uint8)
    n_e = np.zeros(B, np.int64)
    edges = np.zeros((B, epl, 3), np.int32)
    rc = lib().igmc_cpu_extract_batch(C.c_int(nu), C.c_int(nv), _p(u_ptr), _p(u_idx), _p(u_rel), _p(v_ptr), _p(v_idx), _p(lu), _p(lv),
                                      None if li is None else _p(li), C.c_int64(first), C.c_int(B), C.c_int(hop),
                                      C.c_double(sample_ratio), C.c_int(mn), C.c_uint64(seed), C.c_uint64(epoch), C.c_int(cap_u),
                                      C.c_int(cap_v), C.c_int64(epl), _p(n_u), _p(n_v), _p(users), _p(items), _p(ulab), _p(vlab),
                                      _p(n_e), _p(edges))
    assert rc == 0
    if raw:
        return n_u, n_v, users, items, ulab, vlab, n_e, edges
    out = []
    for g in range(B):
        assert n_e[g] >= 0, 'capacity too small for link %d' % g
        us, vs = users[g, :n_u[g]], items[g, :n_v[g]]
        e = edges[g, :n_e[g]]
        tri = sorted((int(us[a]), int(vs[b]), int(r)) for a, b, r in e)
        out.append((us.copy(), vs.copy(), {int(i): int(l) for i, l in zip(us, ulab[g])}, {int(i): int(l) for i, l in zip(vs, vlab[g])},
                    tri))
    return out
