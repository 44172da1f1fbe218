"""Host-side dataset loading: produces the reference's 13-tuple
``(u_features, v_features, adj_train, train_labels, train_u, train_v, val_labels, val_u, val_v,
test_labels, test_u, test_v, class_values)`` whose ``adj_train`` stores rating-label + 1
(reference ``preprocessing.py:190-197, 316-321``) -- the input contract of the hot path.

* :func:`load_data_monti`      -- flixster / douban / yahoo_music, restating the split logic of
  reference ``preprocessing.py:203-333`` on the bundled matrices (read from the ``.npz`` produced by
  ``tests/golden/convert_mat.py``, or from the original MATLAB -v7.3 ``.mat`` through ``igmc_amd/mat73.py``).
* :func:`synth_ml`             -- MovieLens-shaped synthetic generator of SURVEY.md section 8(d)
  (MovieLens itself is not available offline).
* :func:`create_trainvaltest_split` -- random split with the reference's proportions
  (``preprocessing.py:159-197``) over real MovieLens files when present under ``raw_data/``, else over
  :func:`synth_ml`.

One-off host preprocessing (seconds); not part of the accelerated path.
"""
import os

import numpy as np
import scipy.sparse as sp

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ML_HIST = {
    'ml_100k': (943, 1682, 100000, [6110, 11370, 27145, 34174, 21201]),
    'ml_1m': (6040, 3706, 1000209, [56174, 107557, 261197, 348971, 226310]),
    # ML-10M's ten half-star rating levels (levels 1..10 = ratings 0.5..5.0, approximate ML-10M histogram) on ml_1m's graph
    # shape: the R = 10 counterpart of the headline workload (reference Main.py:155-163 lists ml_10m with flixster); the
    # full 71567 x 10681 x 10M graph is a raw_data/ml_10m matter
    'ml_10m_lite': (6040, 3706, 1000209, [94988, 384180, 118278, 790306, 370178, 2356676, 879764, 2875850, 585022, 1544812]),
}


def _find_raw(dataset, fname):
    for base in ('raw_data', os.path.join(_PKG_ROOT, 'raw_data')):
        p = os.path.join(base, dataset, fname)
        if os.path.exists(p):
            return p
    return None


def _load_monti_arrays(dataset):
    """-> dict(shape, M (coo triplets), Otraining, Otest, W_* optional)."""
    p = _find_raw(dataset, 'training_test_dataset.npz')
    if p is not None:
        z = np.load(p)
        return {k: z[k] for k in z.files}
    p = _find_raw(dataset, 'training_test_dataset.mat')
    if p is None:
        raise FileNotFoundError('raw_data/%s/training_test_dataset.{npz,mat} not found' % dataset)
    from . import mat73      # the HDF5 subset of MATLAB's -v7.3 files, read directly (the reference uses h5py)
    out = {}
    db = mat73.File(p)
    def dense(name):
        ds = db[name]
        if isinstance(ds, mat73.Group) and 'ir' in ds.keys():
            return sp.csc_matrix((np.asarray(ds['data']), np.asarray(ds['ir']), np.asarray(ds['jc']))
                                 ).astype(np.float32).toarray()
        return np.asarray(ds).astype(np.float32).T   # MATLAB column-major (ref preprocessing.py:49-51)
    M = dense('M')
    out['shape'] = np.array(M.shape, np.int64)
    r, c = np.nonzero(M)
    out['M_row'], out['M_col'], out['M_val'] = r.astype(np.int32), c.astype(np.int32), M[r, c]
    for k in ('Otraining', 'Otest'):
        r, c = np.nonzero(dense(k))
        out[k + '_row'], out[k + '_col'] = r.astype(np.int32), c.astype(np.int32)
    for k in ('W_users', 'W_movies', 'W_tracks'):
        if k in db.keys():
            W = dense(k)
            r, c = np.nonzero(W)
            out[k + '_shape'] = np.array(W.shape, np.int64)
            out[k + '_row'], out[k + '_col'], out[k + '_val'] = r.astype(np.int32), c.astype(np.int32), W[r, c]
    return out


def _side(z, key, n):
    if key + '_row' in z:
        return sp.csr_matrix((z[key + '_val'], (z[key + '_row'], z[key + '_col'])),
                             shape=tuple(int(x) for x in z[key + '_shape']))
    return sp.identity(n, format='csr')


def load_data_monti(dataset, testing=False, rating_map=None, post_rating_map=None):
    """Monti et al. splits (reference ``preprocessing.py:203-333``)."""
    z = _load_monti_arrays(dataset)
    num_users, num_items = int(z['shape'][0]), int(z['shape'][1])
    u_nodes = z['M_row'].astype(np.int64)          # np.where(M) order = row-major
    v_nodes = z['M_col'].astype(np.int64)
    ratings = z['M_val'].astype(np.float64)
    if rating_map is not None:
        ratings = np.array([rating_map[x] for x in ratings], dtype=np.float64)

    if dataset == 'flixster':
        u_features, v_features = _side(z, 'W_users', num_users), _side(z, 'W_movies', num_items)
    elif dataset == 'douban':
        u_features, v_features = _side(z, 'W_users', num_users), sp.identity(num_items, format='csr')
    elif dataset == 'yahoo_music':
        u_features, v_features = sp.identity(num_users, format='csr'), _side(z, 'W_tracks', num_items)
    else:
        raise ValueError(dataset)

    class_values = np.sort(np.unique(ratings))
    rating_dict = {r: i for i, r in enumerate(class_values.tolist())}
    labels = np.full(num_users * num_items, -1, dtype=np.int32)     # neutral_rating = -1
    labels[u_nodes * num_items + v_nodes] = np.array([rating_dict[r] for r in ratings], dtype=np.int32)

    tr_u, tr_v = z['Otraining_row'].astype(np.int64), z['Otraining_col'].astype(np.int64)
    te_u, te_v = z['Otest_row'].astype(np.int64), z['Otest_col'].astype(np.int64)
    num_train_all = len(tr_u)
    num_test = len(te_u)
    num_val = int(np.ceil(num_train_all * 0.2))
    num_train = num_train_all - num_val

    pairs_train = np.stack([tr_u, tr_v], 1)
    # internal shuffle of the training set before the validation split-off (ref :275-280)
    rand_idx = list(range(num_train_all))
    np.random.seed(42)
    np.random.shuffle(rand_idx)
    pairs_train = pairs_train[rand_idx]
    pairs = np.concatenate([pairs_train, np.stack([te_u, te_v], 1)], 0)
    idx = pairs[:, 0] * num_items + pairs[:, 1]

    val_idx, train_idx, test_idx = idx[:num_val], idx[num_val:num_val + num_train], idx[num_val + num_train:]
    assert len(test_idx) == num_test                                 # ref :289
    val_pairs, train_pairs, test_pairs = pairs[:num_val], pairs[num_val:num_val + num_train], pairs[num_val + num_train:]
    u_test_idx, v_test_idx = test_pairs.T
    u_val_idx, v_val_idx = val_pairs.T
    u_train_idx, v_train_idx = train_pairs.T
    train_labels, val_labels, test_labels = labels[train_idx], labels[val_idx], labels[test_idx]

    if testing:
        u_train_idx = np.hstack([u_train_idx, u_val_idx])
        v_train_idx = np.hstack([v_train_idx, v_val_idx])
        train_labels = np.hstack([train_labels, val_labels])
        train_idx = np.hstack([train_idx, val_idx])

    # training adjacency: values = label + 1  (ref :312-321)
    rating_mx_train = np.zeros(num_users * num_items, dtype=np.float32)
    if post_rating_map is None:
        rating_mx_train[train_idx] = labels[train_idx].astype(np.float32) + 1.
    else:
        rating_mx_train[train_idx] = np.array([post_rating_map[r] for r in class_values[labels[train_idx]]]) + 1.
    rating_mx_train = sp.csr_matrix(rating_mx_train.reshape(num_users, num_items))

    return (sp.csr_matrix(u_features), sp.csr_matrix(v_features), rating_mx_train, train_labels, u_train_idx,
            v_train_idx, val_labels, u_val_idx, v_val_idx, test_labels, u_test_idx, v_test_idx, class_values)


# ------------------------------------------------------------------------------------------ synthetic ML
def synth_ml(n_users, n_items, nnz, hist, seed=0):
    """MovieLens-shaped bipartite rating list.

    user activity ~ LogNormal(0,1), item popularity ~ LogNormal(0,1.6); user degree
    ``clip(round(s*a_u), 20, 0.65*n_items)`` with ``s`` bisected so the degrees sum to ``nnz``; every user
    draws that many distinct items with probability proportional to popularity (Gumbel top-k); ratings iid
    from the real MovieLens histogram.  Returns (u, v, rating_value) in user-major order.
    """
    rng = np.random.default_rng(seed)
    a = rng.lognormal(0.0, 1.0, n_users)
    b = rng.lognormal(0.0, 1.6, n_items)
    lo_d, hi_d = 20, int(0.65 * n_items)

    def degs(s):
        return np.clip(np.round(s * a), lo_d, hi_d).astype(np.int64)
    lo, hi = 0.0, float(nnz)
    for _ in range(100):
        mid = 0.5 * (lo + hi)
        if degs(mid).sum() < nnz:
            lo = mid
        else:
            hi = mid
    d = degs(hi)
    diff = int(d.sum() - nnz)            # fix up to exactly nnz
    order = rng.permutation(n_users)
    k = 0
    while diff != 0:
        u = order[k % n_users]
        k += 1
        if diff > 0 and d[u] > lo_d:
            d[u] -= 1
            diff -= 1
        elif diff < 0 and d[u] < hi_d:
            d[u] += 1
            diff += 1
    logb = np.log(b)
    us, vs = [], []
    chunk = 512
    for s in range(0, n_users, chunk):
        e = min(n_users, s + chunk)
        keys = logb[None, :] + rng.gumbel(size=(e - s, n_items))
        rank = np.argsort(-keys, axis=1)
        for i in range(s, e):
            items = np.sort(rank[i - s, :d[i]])
            us.append(np.full(d[i], i, np.int64))
            vs.append(items.astype(np.int64))
    u = np.concatenate(us)
    v = np.concatenate(vs)
    p = np.asarray(hist, np.float64)
    p = p / p.sum()
    r = rng.choice(np.arange(1, len(hist) + 1), size=len(u), p=p).astype(np.float64)
    return u, v, r


def _py_shuffle_perm(n, seed):
    """The permutation ``random.seed(seed); random.shuffle(rows)`` applies to a list of ``n`` rows (reference
    ``data_utils.py:155-157``: "shuffle here like cf-nade paper with python's own random class"): the Fisher-Yates
    walk depends on the generator and the length only."""
    import random
    idx = list(range(n))
    rnd = random.Random(seed)
    rnd.shuffle(idx)
    return np.asarray(idx, dtype=np.int64)


def _onehot_blocks(columns):
    """One-hot block per column, values numbered in ``np.unique`` order, blocks side by side (reference
    ``data_utils.py:283-301``)."""
    dicts, cntr = [], 0
    for col in columns:
        feats = np.unique(np.asarray(col)).tolist()
        dicts.append({f: i for i, f in enumerate(feats, start=cntr)})
        cntr += len(feats)
    return dicts, cntr


def _maybe_int(values):
    """pandas' type inference for a text column: integers when every field parses as one, else strings."""
    try:
        return [int(x) for x in values]
    except ValueError:
        return list(values)


def _load_real_movielens(dataset, seed=1234):
    """MovieLens raw files an operator dropped into ``raw_data/<dataset>/`` (reference ``data_utils.load_data``,
    ``data_utils.py:88-380``; nothing is downloaded here).  Returns ``(num_users, num_items, u, v, ratings, u_features,
    v_features)`` with the rows in the reference's shuffled order (python ``random`` with ``seed``), ids mapped to
    0..n-1 in sorted order (``map_data``), or ``None`` when the files are absent.

    * ``ml_100k``: ``u.data`` (tab separated) + ``u.item`` (18 genre flags) + ``u.user`` ([age, gender, one-hot
      occupation]);  * ``ml_1m``: ``ratings.dat`` / ``movies.dat`` / ``users.dat`` (``::`` separated; genres one-hot, users
      one-hot gender | age | occupation | zip-code);  * ``ml_10m``: ``ratings.dat`` only.
    Genre (ml_1m) and occupation (ml_100k) columns are numbered in SORTED order -- the reference numbers them by
    iterating a Python set of strings, which differs between interpreter runs."""
    u_features = v_features = None
    if dataset == 'ml_100k':
        p = _find_raw('ml_100k', 'u.data')
        if p is None:
            return None
        raw = np.loadtxt(p, delimiter='\t', dtype=np.float64, ndmin=2)
    elif dataset in ('ml_1m', 'ml_10m'):
        p = _find_raw(dataset, 'ratings.dat')
        if p is None:
            return None
        with open(p) as f:
            raw = np.array([[float(x) for x in line.split('::')] for line in f if line.strip()], dtype=np.float64)
    else:
        return None
    raw = raw[_py_shuffle_perm(len(raw), seed)]
    u_ids, u = np.unique(raw[:, 0].astype(np.int64), return_inverse=True)
    v_ids, v = np.unique(raw[:, 1].astype(np.int64), return_inverse=True)
    num_users, num_items = len(u_ids), len(v_ids)
    upos = {int(i): k for k, i in enumerate(u_ids.tolist())}
    vpos = {int(i): k for k, i in enumerate(v_ids.tolist())}
    base = os.path.dirname(p)
    if dataset == 'ml_100k' and os.path.exists(os.path.join(base, 'u.item')) and os.path.exists(os.path.join(base, 'u.user')):
        v_features = np.zeros((num_items, ML100K_GENRES), dtype=np.float32)
        with open(os.path.join(base, 'u.item'), encoding='latin-1') as f:
            for line in f:
                row = line.rstrip('\n').split('|')
                if len(row) > 6 and int(row[0]) in vpos:
                    v_features[vpos[int(row[0])]] = [float(x) for x in row[6:6 + ML100K_GENRES]]
        with open(os.path.join(base, 'u.user'), encoding='latin-1') as f:
            users = [line.rstrip('\n').split('|') for line in f if line.strip()]
        occ = {o: i for i, o in enumerate(sorted(set(r[3] for r in users)), start=2)}
        u_features = np.zeros((num_users, 2 + len(occ)), dtype=np.float32)
        for r in users:
            k = upos.get(int(r[0]))
            if k is not None:
                u_features[k, 0] = float(r[1])                     # raw age (data_utils.py:207; the official-split
                u_features[k, 1] = {'M': 0., 'F': 1.}[r[2]]        # loader normalises it, this one does not)
                u_features[k, occ[r[3]]] = 1.
    elif dataset == 'ml_1m' and os.path.exists(os.path.join(base, 'movies.dat')) and os.path.exists(os.path.join(base, 'users.dat')):
        with open(os.path.join(base, 'movies.dat'), encoding='latin-1') as f:
            movies = [line.rstrip('\n').split('::') for line in f if line.strip()]
        genres = sorted(set(g for m in movies for g in m[2].split('|')))
        gpos = {g: i for i, g in enumerate(genres)}
        v_features = np.zeros((num_items, len(genres)), dtype=np.float32)
        for m in movies:
            k = vpos.get(int(m[0]))
            if k is not None:
                for g in m[2].split('|'):
                    v_features[k, gpos[g]] = 1.
        with open(os.path.join(base, 'users.dat'), encoding='latin-1') as f:
            users = [line.rstrip('\n').split('::') for line in f if line.strip()]
        cols = [[r[1] for r in users], [int(r[2]) for r in users], [int(r[3]) for r in users],
                _maybe_int([r[4] for r in users])]
        dicts, nf = _onehot_blocks(cols)
        u_features = np.zeros((num_users, nf), dtype=np.float32)
        for j, r in enumerate(users):
            k = upos.get(int(r[0]))
            if k is not None:
                for c in range(4):
                    u_features[k, dicts[c][cols[c][j]]] = 1.
    if u_features is not None:
        u_features, v_features = sp.csr_matrix(u_features), sp.csr_matrix(v_features)
    return num_users, num_items, u.astype(np.int64), v.astype(np.int64), raw[:, 2].astype(np.float64), u_features, v_features


def create_trainvaltest_split(dataset, seed=1234, testing=False, datasplit_path=None, datasplit_from_file=False,
                              verbose=True, rating_map=None, post_rating_map=None, ratio=1.0, synth_seed=0):
    """Random split with the reference's proportions (``preprocessing.py:159-197``): test = ceil(0.1 n),
    val = ceil(0.9*0.05 n), the rest train; ``testing`` merges val into train.  Data = real MovieLens files if
    present, else the synthetic generator (``data`` field of every report says which)."""
    real = _load_real_movielens(dataset, seed)
    u_features = v_features = None
    if real is not None:
        # rows already in the reference's shuffled order (python `random` with `seed`, data_utils.py:155-157)
        num_users, num_items, u_nodes, v_nodes, ratings, u_features, v_features = real
        source = 'real'
    else:
        if dataset not in ML_HIST:
            raise FileNotFoundError('no raw data for %s and no synthetic spec' % dataset)
        num_users, num_items, nnz, hist = ML_HIST[dataset]
        u_nodes, v_nodes, ratings = synth_ml(num_users, num_items, nnz, hist, seed=synth_seed)
        source = 'synthetic'
        # the reference shuffles inside load_data (data_utils.py) with `seed`; same role here
        perm = np.random.default_rng(seed).permutation(len(ratings))
        u_nodes, v_nodes, ratings = u_nodes[perm], v_nodes[perm], ratings[perm]
    if rating_map is not None:
        ratings = np.array([rating_map[x] for x in ratings], dtype=np.float64)
    class_values = np.sort(np.unique(ratings))
    rating_dict = {r: i for i, r in enumerate(class_values.tolist())}
    n = len(ratings)
    num_test = int(np.ceil(n * 0.1))
    num_val = int(np.ceil(n * 0.9 * 0.05))
    num_train = n - num_val - num_test
    all_labels = np.array([rating_dict[r] for r in ratings], dtype=np.int32)
    ntr = int(num_train * ratio)
    u_train_idx, v_train_idx, train_labels = u_nodes[:ntr], v_nodes[:ntr], all_labels[:ntr]
    u_val_idx, v_val_idx, val_labels = (u_nodes[num_train:num_train + num_val], v_nodes[num_train:num_train + num_val],
                                        all_labels[num_train:num_train + num_val])
    u_test_idx, v_test_idx, test_labels = (u_nodes[num_train + num_val:], v_nodes[num_train + num_val:],
                                           all_labels[num_train + num_val:])
    if testing:
        u_train_idx = np.hstack([u_train_idx, u_val_idx])
        v_train_idx = np.hstack([v_train_idx, v_val_idx])
        train_labels = np.hstack([train_labels, val_labels])
    if post_rating_map is None:
        data = train_labels + 1.
    else:
        data = np.array([post_rating_map[r] for r in class_values[train_labels]]) + 1.
    rating_mx_train = sp.csr_matrix((data.astype(np.float32), (u_train_idx, v_train_idx)),
                                    shape=[num_users, num_items], dtype=np.float32)
    if verbose:
        print('%s (%s): %d users, %d items, %d ratings, train %d / val %d / test %d' % (
            dataset, source, num_users, num_items, n, len(train_labels), len(val_labels), len(test_labels)))
    return (u_features, v_features, rating_mx_train, train_labels, u_train_idx, v_train_idx, val_labels, u_val_idx,
            v_val_idx, test_labels, u_test_idx, v_test_idx, class_values)


# ------------------------------------------------------------------------------------------ official ML-100K split
ML100K_GENRES = 18      # u.item columns 6.. ('unknown' at column 5 is skipped, reference preprocessing.py:483)


def _official_arrays(dataset, raw_dir=None, verbose=True):
    """(train rows, test rows, item rows, user rows, source): ``u1.base`` / ``u1.test`` as float arrays
    [user id, item id, rating, timestamp] (reference ``preprocessing.py:357-372``), ``u.item`` / ``u.user`` as lists of
    string fields.  Without the files (no network here) a MovieLens-100K-SHAPED stand-in is generated -- same sizes and
    80 000 / 20 000 proportions as the official u1 split -- and that is said loudly."""
    base = raw_dir
    if base is None:
        p = _find_raw(dataset, 'u1.base')
        base = os.path.dirname(p) if p is not None else None
    if base is not None and os.path.exists(os.path.join(base, 'u1.base')):
        tr = np.loadtxt(os.path.join(base, 'u1.base'), delimiter='\t', dtype=np.float64, ndmin=2)
        te = np.loadtxt(os.path.join(base, 'u1.test'), delimiter='\t', dtype=np.float64, ndmin=2)
        items, users = None, None
        if os.path.exists(os.path.join(base, 'u.item')):
            with open(os.path.join(base, 'u.item'), encoding='latin-1') as f:
                items = [line.rstrip('\n').split('|') for line in f if line.strip()]
        if os.path.exists(os.path.join(base, 'u.user')):
            with open(os.path.join(base, 'u.user'), encoding='latin-1') as f:
                users = [line.rstrip('\n').split('|') for line in f if line.strip()]
        return tr, te, items, users, 'real'
    if dataset not in ML_HIST:
        raise FileNotFoundError('raw_data/%s/u1.base not found and no synthetic spec' % dataset)
    import sys
    sys.stderr.write('igmc_amd.preprocessing: raw_data/%s/{u1.base,u1.test} NOT FOUND -- using the MovieLens-shaped SYNTHETIC '
                     'stand-in (80/20 split like u1); RMSEs are not comparable with published %s numbers\n' % (dataset, dataset))
    num_users, num_items, nnz, hist = ML_HIST[dataset]
    u, v, r = synth_ml(num_users, num_items, nnz, hist, seed=0)
    perm = np.random.default_rng(1).permutation(len(r))
    n_test = len(r) // 5
    rows = np.stack([u + 1, v + 1, r, np.arange(len(r), dtype=np.float64)], 1).astype(np.float64)[perm]
    return rows[n_test:], rows[:n_test], None, None, 'synthetic'


def load_official_trainvaltest_split(dataset, testing=False, rating_map=None, post_rating_map=None, ratio=1.0,
                                     raw_dir=None, verbose=True):
    """MovieLens-100K official split (reference ``preprocessing.py:336-586``; used by reference ``Main.py:236-243`` for
    ``ml_100k``): u1.base -> train (+ validation = the first ceil(0.2 n) of a seed-42 shuffle), u1.test -> test; ids
    mapped to 0..n-1 in sorted order (``data_utils.map_data``); ``ratio`` keeps the earliest ``ratio * n`` training
    ratings by timestamp; ``adj_train`` stores label + 1 (or ``post_rating_map`` + 1).  Side features as the reference:
    users = [age / max age, gender, one-hot occupation], items = 18 genre flags (occupation columns in sorted order --
    the reference numbers them by iterating a Python set, which is not reproducible between runs)."""
    tr, te, items, users, source = _official_arrays(dataset, raw_dir, verbose)
    if ratio < 1.0:
        tr = tr[tr[:, -1].argsort()[:int(ratio * len(tr))]]
    data = np.concatenate([tr, te], axis=0)
    u_raw, v_raw = data[:, 0].astype(np.int32), data[:, 1].astype(np.int32)
    ratings = data[:, 2].astype(np.float32)
    if rating_map is not None:
        ratings = np.array([rating_map[x] for x in ratings], dtype=np.float32)
    u_ids, u_nodes = np.unique(u_raw, return_inverse=True)        # sorted unique ids -> 0..n-1 (map_data)
    v_ids, v_nodes = np.unique(v_raw, return_inverse=True)
    num_users, num_items = len(u_ids), len(v_ids)
    ratings = ratings.astype(np.float64)
    class_values = np.sort(np.unique(ratings))
    lab_of = {r: i for i, r in enumerate(class_values.tolist())}
    labels = np.full((num_users, num_items), -1, dtype=np.int32)
    labels[u_nodes, v_nodes] = np.array([lab_of[r] for r in ratings])          # later duplicates win, like the reference
    labels = labels.reshape(-1)
    num_train, num_test = len(tr), len(te)
    num_val = int(np.ceil(num_train * 0.2))
    num_train -= num_val
    pairs = np.stack([u_nodes, v_nodes], 1).astype(np.int64)
    idx = pairs[:, 0] * num_items + pairs[:, 1]
    n_tv = num_train + num_val
    rand_idx = list(range(n_tv))
    np.random.RandomState(42).shuffle(rand_idx)                    # == np.random.seed(42); np.random.shuffle(list)
    idx = np.concatenate([idx[:n_tv][rand_idx], idx[n_tv:]])
    pairs = np.concatenate([pairs[:n_tv][rand_idx], pairs[n_tv:]])
    val_idx, train_idx, test_idx = idx[:num_val], idx[num_val:n_tv], idx[n_tv:]
    assert len(test_idx) == num_test
    (u_val, v_val), (u_train, v_train), (u_test, v_test) = (pairs[:num_val].T, pairs[num_val:n_tv].T, pairs[n_tv:].T)
    train_labels, val_labels, test_labels = labels[train_idx], labels[val_idx], labels[test_idx]
    if testing:
        u_train, v_train = np.hstack([u_train, u_val]), np.hstack([v_train, v_val])
        train_labels = np.hstack([train_labels, val_labels])
        train_idx = np.hstack([train_idx, val_idx])
    mx = np.zeros(num_users * num_items, dtype=np.float32)
    if post_rating_map is None:
        mx[train_idx] = labels[train_idx].astype(np.float32) + 1.
    else:
        mx[train_idx] = np.array([post_rating_map[r] for r in class_values[labels[train_idx]]]) + 1.
    rating_mx_train = sp.csr_matrix(mx.reshape(num_users, num_items))
    # ---- side features (reference :470-520)
    u_features = v_features = None
    if items is not None:
        vpos = {int(i): k for k, i in enumerate(v_ids.tolist())}
        v_features = np.zeros((num_items, ML100K_GENRES), dtype=np.float32)
        for row in items:
            k = vpos.get(int(row[0]))
            if k is not None:
                v_features[k] = [float(x) for x in row[6:6 + ML100K_GENRES]]
        v_features = sp.csr_matrix(v_features)
    if users is not None:
        upos = {int(i): k for k, i in enumerate(u_ids.tolist())}
        occ = {o: i for i, o in enumerate(sorted(set(row[3] for row in users)), start=2)}
        age_max = max(float(row[1]) for row in users)
        u_features = np.zeros((num_users, 2 + len(occ)), dtype=np.float32)
        for row in users:
            k = upos.get(int(row[0]))
            if k is not None:
                u_features[k, 0] = float(row[1]) / age_max
                u_features[k, 1] = {'M': 0., 'F': 1.}[row[2]]
                u_features[k, occ[row[3]]] = 1.
        u_features = sp.csr_matrix(u_features)
    if verbose:
        print('%s official split (%s): %d users, %d items, train %d / val %d / test %d' % (
            dataset, source, num_users, num_items, len(train_labels), len(val_labels), len(test_labels)))
    return (u_features, v_features, rating_mx_train, train_labels, u_train, v_train, val_labels, u_val, v_val,
            test_labels, u_test, v_test, class_values)
