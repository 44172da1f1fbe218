"""Distribution of the engine's per-hop sampler against the reference's ``random.sample`` (reference
``util_functions.py:222-229``: a uniform k-subset of the fringe, drawn afresh per link, hop, side and epoch).

The engine draws the k candidates with the smallest ``igmc_sample_key(salt, id)`` (``include/igmc_rng.h``), salt keyed by
(dataset seed, epoch, link position, hop distance, side).  Determinism, candidate membership, sizes and the C twin's
bit-equality are checked elsewhere; THIS module checks that the draws are *distributed* like uniform k-subsets:

  singles   inclusion count of every candidate over D draws vs D k / n: chi-square with the exact covariance of a uniform
            k-subset (the counts of one draw sum to k: statistic x (n - 1) / n ~ chi2(n - 1));
  pairs     co-inclusion count of every candidate PAIR vs D k (k - 1) / (n (n - 1)): Pearson statistic, null distribution
            calibrated by the same statistic of numpy's own uniform k-subsets (the cells of one draw are dependent);
  sides     user i and item i of the same draw (same seed / epoch / link, the two sides' salts): joint count vs D p_u p_v;
  epochs    the same candidate in epochs e and e + 1 of the same link: joint count vs D p^2;
  ranks     mean normalised id-rank of the chosen candidates (a lowest-id sampler: k / 2n instead of 1 / 2).

Three capped links on ONE synthetic rating graph: n = 150 / 400 / 2000 candidates a side, k = 100.  The same candidate ids
on both sides (user i and item i exist for every i), so that `sides` can pair them.
"""
import numpy as np
import scipy.sparse as sp

N_CAND = (150, 400, 2000)
K = 100
POS_PER_LINK = 100
SEEDS = (1, 2)
EPOCHS = tuple(range(1, 22))          # 100 positions x 2 seeds x 21 epochs = 4 200 draws per link


def graph():
    """Users / items 0..2 are the three target pairs; user L rated items 3 .. 3 + n_L - 1, item L is rated by users
    3 .. 3 + n_L - 1 (ratings 1..5 by a fixed pattern).  The target pairs themselves are unrated."""
    nmax = max(N_CAND)
    rows, cols, vals = [], [], []
    for L, n in enumerate(N_CAND):
        ids = np.arange(3, 3 + n)
        rows += [np.full(n, L), ids]
        cols += [ids, np.full(n, L)]
        vals += [1 + (ids + L) % 5, 1 + (ids * 3 + L) % 5]
    A = sp.csr_matrix((np.concatenate(vals).astype(np.float32), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(3 + nmax, 3 + nmax))
    P = 3 * POS_PER_LINK
    lu = (np.arange(P) % 3).astype(np.int32)
    return A, lu, lu.copy()


def draws_from_twin(seeds=SEEDS, epochs=EPOCHS):
    """{(seed, epoch): (users[P][...], items[P][...])} through ``oracle/extract_cpu.c`` (bit-equal to the HIP kernels:
    ``tests/test_extract_twin.py``)."""
    from oracle import extract_cpu
    A, lu, lv = graph()
    G = extract_cpu.prepare(A)
    out = {}
    for s in seeds:
        for e in epochs:
            n_u, n_v, users, items = extract_cpu.extract_batch(G, lu, lv, 0, len(lu), hop=1, max_nodes_per_hop=K, seed=s, epoch=e,
                                                               cap_u=K + 1, cap_v=K + 1, raw=True)[:4]
            assert np.all(n_u == K + 1) and np.all(n_v == K + 1)
            out[(s, e)] = (users.copy(), items.copy())
    return out


def draws_from_engine(be, seeds=SEEDS, epochs=EPOCHS):
    """The same draws through the HIP extraction kernels (``be``: parity_checks.GpuBackend / EmuBackend)."""
    from igmc_amd import engine
    A, lu, lv = graph()
    P = len(lu)
    g = engine.Graph(A, device=be.device, lib=be.lib)
    b = engine.Batch(g, max_graphs=P, hop=1, max_nodes_per_hop=K)
    b.set_lean(True)
    dlu, dlv, dly = be.dev(lu), be.dev(lv), be.dev(np.ones(P, np.float32))
    out = {}
    for s in seeds:
        for e in epochs:
            b.extract(be.ptr(dlu), be.ptr(dlv), be.ptr(dly), None, 0, P, sample_ratio=1.0, seed=s, epoch=e)
            be.sync()
            d = b.download()
            users, items = np.zeros((P, K + 1), np.int32), np.zeros((P, K + 1), np.int32)
            for p in range(P):
                lo, hi, nu = d['node_off'][p], d['node_off'][p + 1], d['n_users'][p]
                assert nu == K + 1 and hi - lo == 2 * (K + 1)
                users[p], items[p] = d['node_gid'][lo:lo + nu], d['node_gid'][lo + nu:hi]
            out[(s, e)] = (users, items)
    return out


def inclusion(draws, L, side, seeds=SEEDS, epochs=EPOCHS):
    """bool[D, n_L]: draw d (seed-major, then epoch, then position) chose candidate id 3 + j."""
    n = N_CAND[L]
    rows = []
    for s in seeds:
        for e in epochs:
            arr = draws[(s, e)][side][L::3]            # positions of link L
            assert np.all(arr[:, 0] == L)              # the target first
            X = np.zeros((arr.shape[0], n), bool)
            X[np.arange(arr.shape[0])[:, None], arr[:, 1:] - 3] = True
            rows.append(X)
    X = np.concatenate(rows)
    assert np.all(X.sum(1) == K)
    return X


def uniform_inclusion(D, n, rng):
    """bool[D, n] of D uniform K-subsets (numpy's generator: the reference's random.sample in distribution)."""
    X = np.zeros((D, n), bool)
    idx = np.argpartition(rng.random((D, n)), K, axis=1)[:, :K]
    X[np.arange(D)[:, None], idx] = True
    return X


# ---------------------------------------------------------------------------------------------------------- statistics
def singles_chi2(X):
    """(statistic, dof, p): inclusion counts vs D k / n under the exact covariance of uniform k-subsets."""
    from scipy.stats import chi2
    D, n = X.shape
    p = K / n
    c = X.sum(0).astype(np.float64)
    stat = ((c - D * p) ** 2).sum() / (D * p * (1 - p)) * (n - 1) / n
    return stat, n - 1, float(chi2.sf(stat, n - 1))


def pairs_stat(X):
    """Pearson statistic of the co-inclusion counts of all candidate pairs, and the number of pairs."""
    D, n = X.shape
    q = K * (K - 1) / (n * (n - 1))
    Xf = X.astype(np.float32)
    C = Xf.T @ Xf                                      # exact: counts < 2^24
    iu = np.triu_indices(n, 1)
    c = C[iu].astype(np.float64)
    return float(((c - D * q) ** 2).sum() / (D * q * (1 - q))), len(c)


def joint_stat(X, Y):
    """Pearson statistic of the per-candidate joint counts of two inclusion matrices over the same draws (independent
    under the null: expected D p_x p_y each)."""
    D, n = X.shape
    m = min(n, Y.shape[1])
    px, py = K / X.shape[1], K / Y.shape[1]
    c = (X[:, :m] & Y[:, :m]).sum(0).astype(np.float64)
    e = D * px * py
    return float(((c - e) ** 2).sum() / (e * (1 - px * py))), m


def mean_rank(X):
    """Mean normalised id-rank of the chosen candidates (uniform: 1/2; lowest ids: k / 2n), its null sd."""
    D, n = X.shape
    r = (np.arange(n) + 0.5) / n
    per_draw = (X * r).sum(1) / K
    # variance of the mean of a simple random sample of k of the n ranks, then of the mean over D draws
    var = (r.var() / K) * (n - K) / (n - 1) / D
    return float(per_draw.mean()), float(np.sqrt(var))


def calibrated_z(stat_fn, engine_stat, make_null, reps=10):
    """z-score of the engine's statistic against `reps` replicates of the same statistic from a true uniform sampler."""
    null = np.array([stat_fn(*make_null(r)) for r in range(reps)], np.float64)
    sd = null.std(ddof=1)
    return float((engine_stat - null.mean()) / sd), float(null.mean()), float(sd)


def report(draws, seeds=SEEDS, epochs=EPOCHS, reps=10):
    """All statistics of a set of draws; returns (lines, worst) -- `worst` holds what the tests bound."""
    lines, worst = [], dict(p_min=1.0, z_pairs=0.0, z_sides=0.0, z_epochs=0.0, z_rank=0.0)
    ne = len(epochs)
    for L, n in enumerate(N_CAND):
        Xu, Xv = inclusion(draws, L, 0, seeds, epochs), inclusion(draws, L, 1, seeds, epochs)
        D = Xu.shape[0]
        for side, X in (('users', Xu), ('items', Xv)):
            stat, dof, p = singles_chi2(X)
            mr, sd = mean_rank(X)
            zr = (mr - 0.5) / sd
            rng = np.random.default_rng(1000 + 10 * L + (side == 'items'))
            ps, npairs = pairs_stat(X)
            zp, mu, sdp = calibrated_z(lambda Z: pairs_stat(Z)[0], ps, lambda r: (uniform_inclusion(D, n, rng),), reps)
            lines.append('link %d (%s, n = %d, k = %d, %d draws): singles chi2 %.1f on %d dof, p = %.3f; pairs Pearson %.1f over %d pairs '
                         '(uniform sampler: %.1f +- %.1f, z = %+.2f); mean id-rank %.4f (z = %+.2f)'
                         % (L, side, n, K, D, stat, dof, p, ps, npairs, mu, sdp, zp, mr, zr))
            worst['p_min'] = min(worst['p_min'], p)
            worst['z_pairs'] = max(worst['z_pairs'], abs(zp))
            worst['z_rank'] = max(worst['z_rank'], abs(zr))
        # the two sides of the same draw
        rng = np.random.default_rng(2000 + L)
        js, m = joint_stat(Xu, Xv)
        zs, mu, sdj = calibrated_z(lambda A_, B_: joint_stat(A_, B_)[0], js,
                                   lambda r: (uniform_inclusion(D, n, rng), uniform_inclusion(D, n, rng)), reps)
        lines.append('link %d: users x items of the same draw, joint counts of %d ids: Pearson %.1f (independent uniform: %.1f +- %.1f, '
                     'z = %+.2f)' % (L, m, js, mu, sdj, zs))
        worst['z_sides'] = max(worst['z_sides'], abs(zs))
        # consecutive epochs of the same (seed, position): rows are seed-major, epoch, position.  (Epoch e + 1 is the second
        # member of one pair and the first of the next: the null replicates are paired up the same way.)
        def epoch_pairs(X):
            per = X.shape[0] // (len(seeds) * ne)
            Xs = X.reshape(len(seeds), ne, per, n)
            return Xs[:, :-1].reshape(-1, n), Xs[:, 1:].reshape(-1, n)
        a, b = epoch_pairs(Xu)
        es, m = joint_stat(a, b)
        rng = np.random.default_rng(3000 + L)
        ze, mu, sde = calibrated_z(lambda A_, B_: joint_stat(A_, B_)[0], es, lambda r: epoch_pairs(uniform_inclusion(D, n, rng)), reps)
        lines.append('link %d: users of epochs e and e + 1 of the same link, joint counts: Pearson %.1f (independent uniform: %.1f +- '
                     '%.1f, z = %+.2f)' % (L, es, mu, sde, ze))
        worst['z_epochs'] = max(worst['z_epochs'], abs(ze))
    return lines, worst


def lowest_id_draws(seeds=SEEDS, epochs=EPOCHS):
    """What a sampler that always takes the k lowest candidate ids would return (the negative control of the tests)."""
    P = 3 * POS_PER_LINK
    users = np.zeros((P, K + 1), np.int32)
    users[:, 0] = np.arange(P) % 3
    users[:, 1:] = 3 + np.arange(K)
    return {(s, e): (users, users) for s in seeds for e in epochs}
