"""Randomised differential test of the HIP extraction logic (igmc_amd/csrc/extract.hip on the CPU emulation) against
the oracle restatement of reference ``util_functions.py:208-277`` (oracle/extract_ref.py, itself pinned to the
unmodified reference by tests/test_oracle_golden.py): random bipartite rating graphs with empty rows / columns,
targets that are (and are not) rated, isolated targets, 1 and 2 hops.

* no cap: extraction is RNG-free -> node sets, labels and induced (u, v, relation) triples must match exactly;
* cap / sample_ratio binding: the oracle's ``random.sample`` choices are replayed into the kernels
  (``igmc_extract_batch_replay``), which must emit exactly the induced block of those nodes, and the free-running
  hash sampler must pick the right NUMBER of nodes from the right candidate sets (parity_checks.check_sampled).
"""
import pytest

import parity_checks as PC
from helpers import random_case


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


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
