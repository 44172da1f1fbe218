"""Golden enclosing subgraphs produced by the UNMODIFIED reference extractor.

Run in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden.py

Imports ``/root/reference/util_functions.py`` through the 3-class ``torch_geometric`` stub in
``oracle/ref_stub`` and calls its ``subgraph_extraction_labeling`` (reference
``util_functions.py:208-277``) on:

* ``hand``      the 3x4 known-answer graph of SURVEY.md section 8(c);
* ``flixster`` / ``douban`` / ``yahoo_music``  first train and test links of the bundled data,
  h=1, default cap 10000 (never binds -> RNG-free, reference output is a deterministic golden set);
* ``flixster_h2``  2-hop extraction;
* ``douban_cap20`` / ``synth_cap`` / ``synth_h2_ratio``  cases where the per-hop cap / sample ratio
  BINDS: the reference draws with CPython ``random.sample`` (not reproducible elsewhere), so the
  golden records the node lists it chose; parity of everything downstream is checked by replaying
  those node lists.

The reference function does not return the node lists; they are captured WITHOUT touching
reference code by wrapping its row indexer in a recording proxy (``Arow[u_nodes][:, v_nodes]``
hands both ordered lists to ``__getitem__``, reference ``:236``).

Output: ``tests/golden/extract_golden.npz``.
"""
import os
import random
import sys
import warnings

import numpy as np
import scipy.sparse as ssp

warnings.simplefilter('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_stub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
import util_functions as REF  # noqa: E402  (the unmodified reference module)
from igmc_amd import preprocessing  # noqa: E402


class _RecCSR(ssp.csr_matrix):
    rec = None

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 2 and isinstance(key[1], list):
            _RecCSR.rec['v_nodes'] = list(key[1])
        return ssp.csr_matrix.__getitem__(self, key)


class _RecRows(object):
    """Recording proxy around the reference's SparseRowIndexer."""

    def __init__(self, real):
        self.real = real
        self.shape = real.shape
        self.rec = {}
        self.final = False

    def __getitem__(self, sel):
        out = self.real[sel]
        self.rec['u_nodes_last'] = list(sel)
        m = _RecCSR(out)
        _RecCSR.rec = self.rec
        return m


def run_case(A, links, labels, class_values, h, sample_ratio, mnph, seed):
    Arow = _RecRows(REF.SparseRowIndexer(A))
    Acol = REF.SparseColIndexer(A.tocsc())
    random.seed(seed)
    recs = []
    for (i, j), lab in zip(links, labels):
        Arow.rec.clear()
        u, v, r, node_labels, max_label, y, _ = REF.subgraph_extraction_labeling(
            (i, j), Arow, Acol, h, sample_ratio, mnph, None, None, class_values, lab)
        u_nodes = Arow.rec['u_nodes_last']      # the last Arow[...] call is Arow[u_nodes] (ref :236)
        v_nodes = Arow.rec['v_nodes']
        nu = len(u_nodes)
        assert max_label == 2 * h + 1 and len(node_labels) == nu + len(v_nodes)
        recs.append(dict(u_nodes=np.asarray(u_nodes, np.int32), v_nodes=np.asarray(v_nodes, np.int32),
                         u=np.asarray(u, np.int32), v=np.asarray(v, np.int32), r=np.asarray(r, np.int32),
                         labels=np.asarray(node_labels, np.uint8), y=np.float32(y)))
    return recs


def pack(out, name, A, links, labels, class_values, h, sample_ratio, mnph, recs):
    A = A.tocoo()
    out[name + '/A_row'], out[name + '/A_col'] = A.row.astype(np.int32), A.col.astype(np.int32)
    out[name + '/A_val'] = A.data.astype(np.uint8)
    out[name + '/A_shape'] = np.array(A.shape, np.int64)
    out[name + '/links'] = np.asarray(links, np.int32)
    out[name + '/link_labels'] = np.asarray(labels, np.int32)
    out[name + '/class_values'] = np.asarray(class_values, np.float64)
    out[name + '/params'] = np.array([h, sample_ratio, -1 if mnph is None else mnph], np.float64)
    for key in ('u_nodes', 'v_nodes', 'u', 'v', 'r', 'labels'):
        out[name + '/' + key] = np.concatenate([x[key] for x in recs]) if recs else np.zeros(0)
        out[name + '/' + key + '_off'] = np.cumsum([0] + [len(x[key]) for x in recs]).astype(np.int64)
    out[name + '/y'] = np.array([x['y'] for x in recs], np.float32)


def main():
    out = {}
    # ---- hand graph (known-answer #1)
    A = ssp.csr_matrix(np.array([[1, 2, 0, 5], [0, 3, 4, 0], [2, 0, 0, 1]], dtype=np.float32))
    cv = np.array([1., 2., 3., 4., 5.])
    links = [(0, 1), (1, 2), (2, 3), (0, 0), (1, 1)]
    labels = [1, 3, 0, 0, 2]
    pack(out, 'hand', A, links, labels, cv, 1, 1.0, None, run_case(A, links, labels, cv, 1, 1.0, None, 1))
    pack(out, 'hand_h2', A, links, labels, cv, 2, 1.0, None, run_case(A, links, labels, cv, 2, 1.0, None, 1))

    # ---- bundled datasets (loader restated in igmc_amd.preprocessing; pinned separately)
    os.chdir(ROOT)
    for name, ntr, nte in (('flixster', 24, 24), ('douban', 12, 12), ('yahoo_music', 24, 24)):
        (_, _, adj, tr_l, tr_u, tr_v, _, _, _, te_l, te_u, te_v, cv) = preprocessing.load_data_monti(name, testing=True)
        links = list(zip(tr_u[:ntr].tolist(), tr_v[:ntr].tolist())) + list(zip(te_u[:nte].tolist(), te_v[:nte].tolist()))
        labels = tr_l[:ntr].tolist() + te_l[:nte].tolist()
        # only the rows/cols touched are needed, but the matrix is small: keep it whole for flixster/yahoo
        recs = run_case(adj, links, labels, cv, 1, 1.0, 10000, 1)
        pack(out, name, adj, links, labels, cv, 1, 1.0, 10000, recs)
        if name == 'flixster':
            recs = run_case(adj, links[:6], labels[:6], cv, 2, 1.0, 10000, 1)
            pack(out, 'flixster_h2', adj, links[:6], labels[:6], cv, 2, 1.0, 10000, recs)
        if name == 'douban':
            recs = run_case(adj, links, labels, cv, 1, 1.0, 20, 7)
            pack(out, 'douban_cap20', adj, links, labels, cv, 1, 1.0, 20, recs)

    # ---- small synthetic MovieLens-shaped graph where cap / ratio bind
    u, v, r = preprocessing.synth_ml(300, 200, 9000, preprocessing.ML_HIST['ml_100k'][3], seed=3)
    A = ssp.csr_matrix((r.astype(np.float32), (u, v)), shape=(300, 200))   # values = label + 1 (ratings 1..5)
    cv = np.array([1., 2., 3., 4., 5.])
    rng = np.random.default_rng(5)
    pick = rng.choice(len(u), 16, replace=False)
    links = list(zip(u[pick].tolist(), v[pick].tolist()))
    labels = (r[pick].astype(int) - 1).tolist()
    pack(out, 'synth_cap', A, links, labels, cv, 1, 1.0, 15, run_case(A, links, labels, cv, 1, 1.0, 15, 11))
    pack(out, 'synth_h2_ratio', A, links[:8], labels[:8], cv, 2, 0.5, 12,
         run_case(A, links[:8], labels[:8], cv, 2, 0.5, 12, 13))
    pack(out, 'synth_nocap', A, links, labels, cv, 1, 1.0, None, run_case(A, links, labels, cv, 1, 1.0, None, 1))

    dst = os.path.join(HERE, 'extract_golden.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes;', len(out), 'arrays')
    # known-answer #1: link (0,1) of the hand graph
    o = out
    n0 = slice(o['hand/u_off'][0], o['hand/u_off'][1])
    print('hand (0,1): u', o['hand/u'][n0], 'v', o['hand/v'][n0], 'r', o['hand/r'][n0],
          'labels', o['hand/labels'][o['hand/labels_off'][0]:o['hand/labels_off'][1]])


if __name__ == '__main__':
    main()
