"""Golden outputs of the REFERENCE loader ``preprocessing.load_official_trainvaltest_split`` (reference
``preprocessing.py:336-586``: MovieLens-100K u1.base / u1.test official split + side features) on a small
MovieLens-100K-FORMAT fixture (``tests/golden/ml_100k_mini/{u1.base,u1.test,u.item,u.user}``, generated here with a fixed
seed -- the real files cannot be fetched offline).

Run in the build container with the interpreter that has h5py + pandas:

    /opt/conda/bin/python3.9 /root/repo/tests/golden/make_golden_official.py

``np.float`` (used at reference preprocessing.py:516) was removed from numpy 1.24; it is aliased back for the run -- the
reference module itself is imported unmodified.  Writes ``tests/golden/official_ml_100k_mini.npz``;
``tests/test_oracle_golden.py`` pins ``igmc_amd.preprocessing.load_official_trainvaltest_split`` against it.
"""
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np

warnings.simplefilter('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, 'ml_100k_mini')


def make_fixture():
    rng = np.random.default_rng(7)
    nu, nv = 60, 90
    uid = np.sort(rng.choice(np.arange(1, 200), nu, replace=False))        # non-contiguous ids (map_data)
    vid = np.sort(rng.choice(np.arange(1, 400), nv, replace=False))
    pairs = set()
    while len(pairs) < 2400:
        pairs.add((int(rng.choice(uid)), int(rng.choice(vid))))
    pairs = sorted(pairs)
    order = rng.permutation(len(pairs))
    rows = [(pairs[i][0], pairs[i][1], int(rng.choice([1, 2, 3, 4, 5], p=[.06, .11, .27, .34, .22])),
             int(874724710 + rng.integers(0, 10 ** 7))) for i in order]
    os.makedirs(FIX, exist_ok=True)
    with open(os.path.join(FIX, 'u1.base'), 'w') as f:
        for r in rows[:2000]:
            f.write('%d\t%d\t%d\t%d\n' % r)
    with open(os.path.join(FIX, 'u1.test'), 'w') as f:
        for r in rows[2000:]:
            f.write('%d\t%d\t%d\t%d\n' % r)
    with open(os.path.join(FIX, 'u.item'), 'w', encoding='latin-1') as f:
        for v in vid.tolist() + [999]:                                       # one movie that was never rated
            g = rng.integers(0, 2, 19)
            f.write('%d|Movie %d (1995)|01-Jan-1995||http://x/%d|%s\n' % (v, v, v, '|'.join(str(int(x)) for x in g)))
    occ = ['artist', 'doctor', 'educator', 'engineer', 'none', 'student', 'writer']
    with open(os.path.join(FIX, 'u.user'), 'w') as f:
        for u in uid.tolist() + [998]:
            f.write('%d|%d|%s|%s|%05d\n' % (u, int(rng.integers(12, 70)), rng.choice(['M', 'F']), rng.choice(occ),
                                           int(rng.integers(0, 99999))))


def main():
    if not os.path.exists(os.path.join(FIX, 'u1.base')):
        make_fixture()
    np.float = float                           # removed from numpy >= 1.24; reference preprocessing.py:516 uses it
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'raw_data'))
    shutil.copytree(FIX, os.path.join(tmp, 'raw_data', 'ml_100k'))
    os.chdir(tmp)
    sys.path.insert(0, '/root/reference')
    import preprocessing as P                  # the unmodified reference module
    rec = {}
    rm5 = None
    for tag, kw in (('T', dict(testing=True)), ('F', dict(testing=False)), ('R', dict(testing=True, ratio=0.5)),
                    ('M', dict(testing=True, post_rating_map={1.: 0, 2.: 0, 3.: 1, 4.: 2, 5.: 2}))):
        o = P.load_official_trainvaltest_split('ml_100k', **kw)
        A = o[2].tocoo()
        order = np.lexsort((A.col, A.row))
        rec['adj_row_' + tag] = A.row[order].astype(np.int32)
        rec['adj_col_' + tag] = A.col[order].astype(np.int32)
        rec['adj_val_' + tag] = A.data[order].astype(np.uint8)
        rec['adj_shape_' + tag] = np.array(A.shape)
        for k, i in (('train_labels', 3), ('train_u', 4), ('train_v', 5), ('val_labels', 6), ('val_u', 7),
                     ('val_v', 8), ('test_labels', 9), ('test_u', 10), ('test_v', 11)):
            rec[k + '_' + tag] = np.asarray(o[i]).astype(np.int32)
        rec['class_values'] = np.asarray(o[12], dtype=np.float64)
        if tag == 'T':
            rec['v_features'] = np.asarray(o[1].todense(), dtype=np.float32)
            uf = np.asarray(o[0].todense(), dtype=np.float32)
            # occupation columns are numbered by iterating a Python set of strings (hash order differs between
            # interpreter runs): pin age / gender exactly, the occupation block up to a column permutation
            rec['u_features_age_gender'] = uf[:, :2]
            rec['u_features_occ_sorted'] = np.sort(uf[:, 2:], axis=1)
            rec['u_features_occ_colsum_sorted'] = np.sort(uf[:, 2:].sum(0))
    np.savez_compressed(os.path.join(HERE, 'official_ml_100k_mini.npz'), **rec)
    print({k: v.shape for k, v in rec.items()})


if __name__ == '__main__':
    main()
