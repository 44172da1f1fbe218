"""Randomised differential test of the HIP extraction logic (igmc_amd/csrc/extract.hip on the CPU emulation) against
the oracle restatement of reference ``util_functions.py:208-277`` (oracle/extract_ref.py, itself pinned to the
unmodified reference by tests/test_oracle_golden.py): random bipartite rating graphs with empty rows / columns,
targets that are (and are not) rated, isolated targets, 1 and 2 hops.

* no cap: extraction is RNG-free -> node sets, labels and induced (u, v, relation) triples must match exactly;
* cap / sample_ratio binding: the oracle's ``random.sample`` choices are replayed into the kernels
  (``igmc_extract_batch_replay``), which must emit exactly the induced block of those nodes, and the free-running
  hash sampler must pick the right NUMBER of nodes from the right candidate sets (parity_checks.check_sampled).
"""
import random

import numpy as np
import pytest
import scipy.sparse as ssp

import parity_checks as PC
from oracle import extract_ref as X


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


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
    random.seed(seed)
    recs = []
    for (i, j), lab in zip(links, labels):
        u, v, r, node_labels, _, y, _, (un, vn) = X.subgraph_extraction_labeling(
            (i, j), A, Acsc, h, ratio, mnph, None, None, class_values, lab)
        recs.append(dict(u_nodes=np.asarray(un, np.int64), v_nodes=np.asarray(vn, np.int64), u=u, v=v, r=r,
                         labels=np.asarray(node_labels, np.int64), y=float(y)))
    return dict(A=A, links=np.asarray(links, np.int64), link_labels=np.asarray(labels, np.int64),
                class_values=class_values, h=h, sample_ratio=ratio, mnph=mnph, recs=recs)


@pytest.mark.parametrize('h', [1, 2])
def test_uncapped_free_run_matches_oracle(be, h):
    for seed in range(40):
        case = random_case(1000 * h + seed, h)
        _, _, d = PC.extract_case(be, case, replay=False)
        PC.check_against_golden(d, case)


@pytest.mark.parametrize('h,mnph,ratio', [(1, 3, 1.0), (1, 8, 1.0), (2, 4, 1.0), (1, None, 0.5), (2, 5, 0.7)])
def test_capped_replay_and_sampler(be, h, mnph, ratio):
    for seed in range(12):
        case = random_case(7000 + 100 * h + seed, h, mnph=mnph, ratio=ratio)
        _, _, d = PC.extract_case(be, case, replay=True)
        PC.check_against_golden(d, case)
        if ratio == 1.0:            # int(ratio * len) of a DIFFERENT random subset need not give the reference's sizes
            _, _, d2 = PC.extract_case(be, case, replay=False, seed=seed, epoch=1)
            PC.check_sampled(d2, case)
