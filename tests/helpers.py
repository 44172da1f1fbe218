"""Shared helpers for the test-suite (host side only)."""
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as ssp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def emu_lib():
    """Host emulation build of the HIP sources (kernel-logic tests only; never the product path)."""
    from igmc_amd import _lib, build
    # IGMC_EMU_LIB: a sanitizer build of the same sources (tools/sanitize_emu.sh runs part of the suite under UBSan)
    path = os.environ.get('IGMC_EMU_LIB') or build.build_emu()
    return _lib.bind(ctypes.CDLL(path), path)


def hand_graph():
    """Known-answer graph #1 of SURVEY.md section 8(c)."""
    return ssp.csr_matrix(np.array([[1, 2, 0, 5], [0, 3, 4, 0], [2, 0, 0, 1]], dtype=np.float32))


def random_rating_graph(n_users, n_items, density, n_rel, seed):
    rng = np.random.default_rng(seed)
    mask = rng.random((n_users, n_items)) < density
    vals = rng.integers(1, n_rel + 1, size=(n_users, n_items))
    return ssp.csr_matrix((mask * vals).astype(np.float32))


def batch_to_pyg(d, num_labels):
    """Downloaded engine batch -> PyG-style arrays (x one-hot, edge_index [src;dst], edge_type, batch, y)."""
    import torch
    N = d['N']
    dst = np.repeat(np.arange(N, dtype=np.int64), np.diff(d['row_ptr']).astype(np.int64))
    src = d['col'].astype(np.int64)
    x = np.zeros((N, num_labels), np.float32)
    x[np.arange(N), d['node_label']] = 1.0

    class B(object):
        pass
    b = B()
    b.x = torch.from_numpy(x)
    b.edge_index = torch.from_numpy(np.stack([src, dst], 0))
    b.edge_type = torch.from_numpy(d['erel'].astype(np.int64))
    b.batch = torch.from_numpy(d['node_graph'].astype(np.int64))
    b.y = torch.from_numpy(d['y'].copy())
    b.num_graphs = d['B']
    return b


def graph_canonical(d, g):
    """Order-independent description of graph g of a downloaded batch:
    (sorted user ids, sorted item ids, {gid: label}, sorted (u_gid, v_gid, rel) triples of the
    user->item... both directions checked for symmetry)."""
    lo, hi = d['node_off'][g], d['node_off'][g + 1]
    nu = d['n_users'][g]
    gid = d['node_gid']
    lab = d['node_label']
    users = gid[lo:lo + nu]
    items = gid[lo + nu:hi]
    fwd, bwd = [], []
    for i in range(lo, hi):
        for p in range(d['row_ptr'][i], d['row_ptr'][i + 1]):
            c = d['col'][p]
            assert lo <= c < hi, 'edge leaves its graph'
            if i < lo + nu:      # dst is a user, src must be an item
                assert c >= lo + nu
                bwd.append((gid[i], gid[c], int(d['erel'][p])))
            else:
                assert c < lo + nu
                fwd.append((gid[c], gid[i], int(d['erel'][p])))
            assert d['elab'][p] == lab[c]
    fwd.sort()
    bwd.sort()
    assert fwd == bwd, 'CSR is not symmetric'
    ulab = {int(gid[i]): int(lab[i]) for i in range(lo, lo + nu)}
    vlab = {int(gid[i]): int(lab[i]) for i in range(lo + nu, hi)}
    return users, items, ulab, vlab, np.array(fwd, dtype=np.int64).reshape(-1, 3)


# ------------------------------------------------------------------ golden fixtures
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_extract_golden():
    """-> {case: dict(A csr, links, link_labels, class_values, h, sample_ratio, mnph, recs=[...])}"""
    z = np.load(os.path.join(GOLDEN, 'extract_golden.npz'))
    names = sorted(set(k.split('/')[0] for k in z.files))
    out = {}
    for n in names:
        g = lambda k: z[n + '/' + k]
        shape = tuple(int(x) for x in g('A_shape'))
        A = ssp.csr_matrix((g('A_val').astype(np.float32), (g('A_row'), g('A_col'))), shape=shape)
        h, ratio, mnph = g('params')
        recs = []
        for i in range(len(g('y'))):
            rec = {}
            for key in ('u_nodes', 'v_nodes', 'u', 'v', 'r', 'labels'):
                off = g(key + '_off')
                rec[key] = g(key)[off[i]:off[i + 1]]
            rec['y'] = float(g('y')[i])
            recs.append(rec)
        out[n] = dict(A=A, links=g('links'), link_labels=g('link_labels'), class_values=g('class_values'),
                      h=int(h), sample_ratio=float(ratio), mnph=None if mnph < 0 else int(mnph), recs=recs)
    return out


def golden_canonical(rec):
    """Reference record -> (user ids, item ids, {uid: label}, {vid: label}, sorted (u,v,rel) triples)."""
    un, vn = rec['u_nodes'], rec['v_nodes']
    nu = len(un)
    ulab = {int(g): int(l) for g, l in zip(un, rec['labels'][:nu])}
    vlab = {int(g): int(l) for g, l in zip(vn, rec['labels'][nu:])}
    t = np.stack([un[rec['u']], vn[rec['v'] - nu], rec['r']], 1).astype(np.int64) if len(rec['u']) else np.zeros((0, 3), np.int64)
    if len(t):
        t = t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]
    return un, vn, ulab, vlab, t


# ------------------------------------------------------------------ random cases (oracle-generated records)
def random_case(seed, h, mnph=None, ratio=1.0, n_links=6):
    rng = np.random.default_rng(seed)
    nu, nv = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    n_rel = int(rng.integers(2, 7))
    dens = float(rng.choice([0.03, 0.1, 0.3, 0.6]))
    mask = rng.random((nu, nv)) < dens
    mask[rng.integers(0, nu)] = False                       # an empty user row
    mask[:, rng.integers(0, nv)] = False                    # an empty item column
    vals = rng.integers(1, n_rel + 1, size=(nu, nv))
    A = ssp.csr_matrix((mask * vals).astype(np.float32))
    A.eliminate_zeros()
    Acsc = A.tocsc()
    links, labels = [], []
    rows, cols = A.nonzero()
    for k in range(n_links):
        if k % 2 == 0 and len(rows):                        # a rated pair (training link: its entry is removed)
            p = int(rng.integers(0, len(rows)))
            links.append((int(rows[p]), int(cols[p])))
        else:                                               # any pair (test link; may be isolated)
            links.append((int(rng.integers(0, nu)), int(rng.integers(0, nv))))
        labels.append(int(rng.integers(0, n_rel)))
    class_values = np.arange(1, n_rel + 1, dtype=np.float64)
    import random
    from oracle import extract_ref as X
    random.seed(seed)
    recs = []
    for (i, j), lab in zip(links, labels):
        u, v, r, node_labels, _, y, _, (un, vn) = X.subgraph_extraction_labeling(
            (i, j), A, Acsc, h, ratio, mnph, None, None, class_values, lab)
        recs.append(dict(u_nodes=np.asarray(un, np.int64), v_nodes=np.asarray(vn, np.int64), u=u, v=v, r=r,
                         labels=np.asarray(node_labels, np.int64), y=float(y)))
    return dict(A=A, links=np.asarray(links, np.int64), link_labels=np.asarray(labels, np.int64),
                class_values=class_values, h=h, sample_ratio=ratio, mnph=mnph, recs=recs)




# ------------------------------------------------------------------ model goldens (reference models.py / train_eval.py)
class ModelGolden(object):
    """``tests/golden/model_golden.npz`` (outputs of the UNMODIFIED reference ``models.py`` / ``train_eval.py``,
    ``tests/golden/make_model_golden.py``) for one case."""

    def __init__(self, z, case):
        self.z, self.case = z, case
        if (case + '/ctor') not in z.files:         # geometry-only entry (the headline case's node lists)
            return
        self.R, self.n_side, self.multiply_by, self.force_undirected = [float(x) for x in z[case + '/ctor']]
        self.R, self.n_side, self.force_undirected = int(self.R), int(self.n_side), bool(self.force_undirected)
        self.ARR, self.lr, self.p_edge = [float(x) for x in z[case + '/train/hyper']]

    def _key(self, key):
        k = self.case + '/' + key
        if k not in self.z.files and self.case.startswith('headline_'):      # geometry of the two headline runs: shared
            k = 'headline/' + key
        return k

    def __getitem__(self, key):
        return self.z[self._key(key)]

    def has(self, key):
        return self._key(key) in self.z.files

    def state(self, prefix):
        import torch
        p = self.case + '/' + prefix + '/'
        return {k[len(p):]: torch.from_numpy(self.z[k].copy()) for k in self.z.files if k.startswith(p)}

    def n_steps(self):
        return sum(1 for k in self.z.files if k.startswith(self.case + '/train/out/'))

    def lin_mask(self, s, shape):
        n = int(np.prod(shape))
        return np.unpackbits(self['train/lin_mask/%d' % s])[:n].astype(bool).reshape(shape)

    def edge_mask(self, s):
        if not self.has('train/edge_mask/%d' % s):
            return None
        return np.unpackbits(self['train/edge_mask/%d' % s])[:int(self['train/edge_mask_n/%d' % s])].astype(bool)

    def batch(self, b, num_labels=4):
        """Collated batch ``b`` as the reference's DataLoader produced it."""
        import torch
        p = 'batch%d/' % b
        lab = self[p + 'label'].astype(np.int64)
        x = np.zeros((len(lab), num_labels), np.float32)
        x[np.arange(len(lab)), lab] = 1.0

        class B(object):
            pass
        o = B()
        o.x = torch.from_numpy(x)
        o.edge_index = torch.from_numpy(self[p + 'edge_index'].astype(np.int64))
        o.edge_type = torch.from_numpy(self[p + 'edge_type'].astype(np.int64))
        sizes = self[p + 'sizes'].astype(np.int64)
        o.batch = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes))
        o.y = torch.from_numpy(self[p + 'y'].copy())
        o.num_graphs = len(sizes)
        if self.has(p + 'u_feature'):
            o.u_feature = torch.from_numpy(self[p + 'u_feature'].copy())
            o.v_feature = torch.from_numpy(self[p + 'v_feature'].copy())
        return o


def load_model_golden(case):
    return ModelGolden(np.load(os.path.join(GOLDEN, 'model_golden.npz')), case)


def headline_golden_case(mg):
    """The headline fixture as an extraction case: graph re-synthesised (fingerprint checked), links, node lists."""
    import hashlib
    from igmc_amd import preprocessing
    split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    A = ssp.csr_matrix(split[2])
    A.sort_indices()
    h = hashlib.sha256()
    for a in (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    fp = np.frombuffer(h.digest()[:8], np.uint64)[0]
    assert fp == mg['graph_fingerprint'], 'the synthetic ml_1m graph is not the one the fixture was generated on'
    uo, vo = mg['u_off'], mg['v_off']
    recs = []
    for g in range(len(uo) - 1):
        un, vn = mg['u_nodes'][uo[g]:uo[g + 1]].astype(np.int64), mg['v_nodes'][vo[g]:vo[g + 1]].astype(np.int64)
        labels = np.concatenate([np.where(np.arange(len(un)) == 0, 0, 2), np.where(np.arange(len(vn)) == 0, 1, 3)])
        recs.append(dict(u_nodes=un, v_nodes=vn, labels=labels.astype(np.int64)))
    return dict(A=split[2], links=mg['links'].astype(np.int64), link_labels=mg['link_labels'].astype(np.int64),
                class_values=mg['class_values'], h=1, sample_ratio=1.0, mnph=100, recs=recs)
