"""Convert the reference's bundled Monti ``.mat`` (HDF5 v7.3) files to sparse ``.npz``.

Run ONCE in the build container with the interpreter that has h5py:

    /opt/conda/bin/python3.9 tests/golden/convert_mat.py

Follows the read convention of ``/root/reference/preprocessing.py:32-55``
(dense datasets are transposed on read).  Only the rating matrix ``M`` and the
split masks ``Otraining`` / ``Otest`` are kept, as COO triplets, so the result is
a few hundred KB and can travel to the GPU box (``/root/reference`` cannot).
Side-information graphs (W_users, ...) are kept as COO as well (row-normalised
features are only needed by --use-features).
"""
import os
import sys
import numpy as np
import h5py

REF = '/root/reference/raw_data'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'raw_data')


def load_field(db, name):
    ds = db[name]
    if isinstance(ds, h5py.Group) and 'ir' in ds.keys():
        import scipy.sparse as sp
        data = np.asarray(ds['data']); ir = np.asarray(ds['ir']); jc = np.asarray(ds['jc'])
        return sp.csc_matrix((data, ir, jc)).astype(np.float32).toarray()
    return np.asarray(ds).astype(np.float32).T


def coo(mat):
    r, c = np.nonzero(mat)
    return r.astype(np.int32), c.astype(np.int32), mat[r, c].astype(np.float32)


for name in ['flixster', 'douban', 'yahoo_music']:
    path = os.path.join(REF, name, 'training_test_dataset.mat')
    db = h5py.File(path, 'r')
    out = {}
    M = load_field(db, 'M')
    out['shape'] = np.array(M.shape, dtype=np.int64)
    out['M_row'], out['M_col'], out['M_val'] = coo(M)
    for k in ['Otraining', 'Otest']:
        O = load_field(db, k)
        r, c, _ = coo(O)
        out[k + '_row'], out[k + '_col'] = r, c
    for k in ['W_users', 'W_movies', 'W_tracks']:
        if k in db.keys():
            W = load_field(db, k)
            out[k + '_shape'] = np.array(W.shape, dtype=np.int64)
            out[k + '_row'], out[k + '_col'], out[k + '_val'] = coo(W)
    db.close()
    os.makedirs(os.path.join(OUT, name), exist_ok=True)
    dst = os.path.join(OUT, name, 'training_test_dataset.npz')
    np.savez_compressed(dst, **out)
    print(name, M.shape, 'nnz', len(out['M_val']), 'train', len(out['Otraining_row']),
          'test', len(out['Otest_row']), '->', dst, os.path.getsize(dst))
