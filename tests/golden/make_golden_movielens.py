"""Golden outputs of the REFERENCE random-split loader ``preprocessing.create_trainvaltest_split`` -> ``data_utils.load_data``
(reference ``preprocessing.py:102-197``, ``data_utils.py:88-380``) on small MovieLens-FORMAT fixtures: ``ml_1m`` (``::``
separated ratings.dat / movies.dat / users.dat, generated here with a fixed seed) and ``ml_100k`` (u.data = u1.base +
u1.test of ``tests/golden/ml_100k_mini`` + its u.item / u.user).  The real files cannot be fetched offline.

Run in the build container with the interpreter that has pandas:

    /opt/conda/bin/python3.9 /root/repo/tests/golden/make_golden_movielens.py

Writes ``tests/golden/movielens_mini.npz``; ``tests/test_oracle_golden.py`` pins
``igmc_amd.preprocessing.create_trainvaltest_split`` (real-file branch) against it.  The reference numbers genre / occupation
columns by iterating Python sets of strings (hash order differs between interpreter runs): those blocks are pinned up to a
column permutation, everything else exactly.
"""
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np

warnings.simplefilter('ignore')
HERE = os.path.dirname(os.path.abspath(__file__))
FIX1M = os.path.join(HERE, 'ml_1m_mini')
FIX100K = os.path.join(HERE, 'ml_100k_mini')
GENRES = ['Action', 'Adventure', 'Animation', "Children's", 'Comedy', 'Crime', 'Drama', 'Horror', 'Sci-Fi', 'War']


def make_fixture_1m():
    rng = np.random.default_rng(11)
    nu, nv = 70, 110
    uid = np.sort(rng.choice(np.arange(1, 300), nu, replace=False))        # non-contiguous ids (map_data)
    vid = np.sort(rng.choice(np.arange(1, 900), nv, replace=False))
    pairs = set()
    while len(pairs) < 3000:
        pairs.add((int(rng.choice(uid)), int(rng.choice(vid))))
    pairs = sorted(pairs)
    order = rng.permutation(len(pairs))
    os.makedirs(FIX1M, exist_ok=True)
    with open(os.path.join(FIX1M, 'ratings.dat'), 'w') as f:
        for i in order:
            f.write('%d::%d::%d::%d\n' % (pairs[i][0], pairs[i][1],
                                         int(rng.choice([1, 2, 3, 4, 5], p=[.06, .11, .26, .35, .22])),
                                         int(956700000 + rng.integers(0, 10 ** 7))))
    with open(os.path.join(FIX1M, 'movies.dat'), 'w', encoding='latin-1') as f:
        for v in vid.tolist() + [3952]:                                      # one movie that was never rated
            gs = sorted(set(rng.choice(GENRES, size=int(rng.integers(1, 4))).tolist()))
            f.write('%d::Movie %d (1995)::%s\n' % (v, v, '|'.join(gs)))
    with open(os.path.join(FIX1M, 'users.dat'), 'w') as f:
        for u in uid.tolist() + [6040]:
            f.write('%d::%s::%d::%d::%05d\n' % (u, rng.choice(['M', 'F']), int(rng.choice([1, 18, 25, 35, 45, 50, 56])),
                                               int(rng.integers(0, 21)), int(rng.integers(0, 99999))))


def canon_cols(m):
    """columns in a canonical order (lexicographic on their 0/1 patterns): equal up to a column permutation"""
    m = np.asarray(m, dtype=np.float32)
    return m[:, np.lexsort(m[::-1])]


def record(rec, name, o):
    A = o[2].tocoo()
    order = np.lexsort((A.col, A.row))
    rec[name + '_adj_row'] = A.row[order].astype(np.int32)
    rec[name + '_adj_col'] = A.col[order].astype(np.int32)
    rec[name + '_adj_val'] = A.data[order].astype(np.uint8)
    rec[name + '_adj_shape'] = np.array(A.shape)
    for k, i in (('train_labels', 3), ('train_u', 4), ('train_v', 5), ('val_labels', 6), ('val_u', 7), ('val_v', 8),
                 ('test_labels', 9), ('test_u', 10), ('test_v', 11)):
        rec[name + '_' + k] = np.asarray(o[i]).astype(np.int32)
    rec[name + '_class_values'] = np.asarray(o[12], dtype=np.float64)


def main():
    if not os.path.exists(os.path.join(FIX1M, 'ratings.dat')):
        make_fixture_1m()
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'raw_data'))
    shutil.copytree(FIX1M, os.path.join(tmp, 'raw_data', 'ml_1m'))
    shutil.copytree(FIX100K, os.path.join(tmp, 'raw_data', 'ml_100k'))
    with open(os.path.join(tmp, 'raw_data', 'ml_100k', 'u.data'), 'w') as f:     # the un-split file of the same ratings
        for part in ('u1.base', 'u1.test'):
            f.write(open(os.path.join(FIX100K, part)).read())
    os.chdir(tmp)
    sys.path.insert(0, '/root/reference')
    import preprocessing as P                  # the unmodified reference module
    rec = {}
    for ds in ('ml_1m', 'ml_100k'):
        for tag, kw in (('T', dict(testing=True)), ('F', dict(testing=False)), ('R', dict(testing=True, ratio=0.4)),
                        ('S', dict(testing=True, seed=7))):
            path = os.path.join(tmp, '%s_%s.pkl' % (ds, tag))
            o = P.create_trainvaltest_split(ds, kw.pop('seed', 1234), datasplit_path=path, verbose=False, **kw)
            record(rec, '%s_%s' % (ds, tag), o)
            if tag == 'T':
                uf, vf = np.asarray(o[0].todense(), np.float32), np.asarray(o[1].todense(), np.float32)
                if ds == 'ml_1m':            # users: one-hot gender | age | occupation | zip (np.unique order: exact)
                    rec[ds + '_u_features'] = uf
                    rec[ds + '_v_features_canon'] = canon_cols(vf)          # genres: set order
                else:                        # ml_100k: [age, gender, occupation (set order)], genres in file order
                    rec[ds + '_u_features_age_gender'] = uf[:, :2]
                    rec[ds + '_u_features_occ_canon'] = canon_cols(uf[:, 2:])
                    rec[ds + '_v_features'] = vf
    np.savez_compressed(os.path.join(HERE, 'movielens_mini.npz'), **rec)
    print({k: v.shape for k, v in rec.items() if k.endswith('_T_train_u') or 'features' in k})


if __name__ == '__main__':
    main()
