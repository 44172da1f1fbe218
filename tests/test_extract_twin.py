"""``oracle/extract_cpu.c`` -- an independent OpenMP host implementation of the engine's free-running extraction (SURVEY.md 8(b):
the ``igmc_cpu_*`` twins) -- against the HIP extraction kernels: same node sets, same order, same labels, same induced edges,
for the same (seed, epoch, link position) draws.  CPU: the kernels on the emulator, every golden case incl. per-hop caps,
``sample_ratio`` < 1 and two hops.  GPU: the headline shape (ml_1m-shaped graph, cap 100, batch 50)."""
import numpy as np
import pytest

import parity_checks as PC
from helpers import golden_canonical, graph_canonical, load_extract_golden
from oracle import extract_cpu

CASES = load_extract_golden()


def compare(d, case, twin):
    for gi in range(len(twin)):
        users, items, ulab, vlab, edges = graph_canonical(d, gi)
        tu, tv, tul, tvl, te = twin[gi]
        assert np.array_equal(users, tu) and np.array_equal(items, tv), 'node sets / order of link %d' % gi
        assert ulab == tul and vlab == tvl, 'labels of link %d' % gi
        assert [tuple(int(x) for x in e) for e in edges] == te, 'induced edges of link %d' % gi


@pytest.mark.parametrize('name', sorted(CASES))
def test_twin_matches_the_hip_extraction_on_the_emulator(name):
    be = PC.EmuBackend()
    case = dict(CASES[name])
    n = min(8, len(case['recs']))
    case['recs'], case['links'], case['link_labels'] = case['recs'][:n], case['links'][:n], case['link_labels'][:n]
    for seed, epoch in ((5, 1), (9, 3)):
        _, _, d = PC.extract_case(be, case, replay=False, seed=seed, epoch=epoch)
        twin = extract_cpu.extract_batch(case['A'], case['links'][:, 0], case['links'][:, 1], 0, n, hop=case['h'],
                                         sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'], seed=seed, epoch=epoch)
        compare(d, case, twin)
    # another epoch draws other samples where a cap or a ratio binds (and the same ones where nothing is sampled)
    a = extract_cpu.extract_batch(case['A'], case['links'][:, 0], case['links'][:, 1], 0, n, hop=case['h'],
                                  sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'], seed=5, epoch=2)
    b = extract_cpu.extract_batch(case['A'], case['links'][:, 0], case['links'][:, 1], 0, n, hop=case['h'],
                                  sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'], seed=5, epoch=1)
    differs = any(not np.array_equal(x[0], y[0]) or not np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    if name in ('synth_cap', 'douban_cap20', 'synth_h2_ratio'):
        assert differs


def test_twin_reports_a_capacity_that_is_too_small():
    case = CASES['synth_nocap']
    out = extract_cpu.extract_batch(case['A'], case['links'][:2, 0], case['links'][:2, 1], 0, 2, cap_u=2, cap_v=2, raw=True)
    assert np.all(out[6] == -1)


@pytest.mark.gpu
def test_twin_matches_the_gpu_extraction_at_the_headline_shape():
    import test_gpu_headline as H
    be = PC.GpuBackend()
    case = H.first(H.ml_case('ml_1m', 100, 50, seed=1), 50)
    _, _, d = PC.extract_case(be, case, replay=False, seed=11, epoch=4)
    twin = extract_cpu.extract_batch(case['A'], case['links'][:, 0], case['links'][:, 1], 0, 50, hop=1,
                                     sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'], seed=11, epoch=4)
    compare(d, case, twin)
    assert max(len(t[0]) for t in twin) == 101 and max(len(t[1]) for t in twin) == 101      # (the cap binds on both sides)


@pytest.mark.parametrize('h', [1, 2])
def test_twin_matches_the_oracle_extractor_on_random_graphs(h):
    """No engine in the loop: without caps nothing is sampled, so the twin must return the node sets, labels and induced edges
    of ``oracle/extract_ref.py`` (pinned to the reference) -- empty rows / columns, isolated and rated target pairs."""
    from helpers import random_case
    for seed in range(40):
        case = random_case(5000 * h + seed, h)
        n = len(case['links'])
        twin = extract_cpu.extract_batch(case['A'], case['links'][:, 0], case['links'][:, 1], 0, n, hop=h)
        for (tu, tv, tul, tvl, te), rec, (i, j) in zip(twin, case['recs'], case['links']):
            gun, gvn, gulab, gvlab, gt = golden_canonical(rec)
            assert tu[0] == i and tv[0] == j and sorted(tu) == sorted(gun.tolist()) and sorted(tv) == sorted(gvn.tolist())
            assert tul == gulab and tvl == gvlab
            assert te == [tuple(int(x) for x in e) for e in gt]


@pytest.mark.parametrize('name', sorted(CASES))
def test_twin_matches_the_reference_goldens_where_nothing_is_sampled(name):
    """Golden records of the UNMODIFIED reference (tests/golden/make_golden.py): wherever neither the per-hop cap nor the
    ratio cut a candidate set, the twin's subgraph IS the reference's."""
    case = CASES[name]
    n = min(8, len(case['recs']))
    args = (case['A'], case['links'][:, 0], case['links'][:, 1], 0, n)
    free = extract_cpu.extract_batch(*args, hop=case['h'])
    capped = extract_cpu.extract_batch(*args, hop=case['h'], sample_ratio=case['sample_ratio'], max_nodes_per_hop=case['mnph'])
    compared = 0
    for g in range(n):
        if not (np.array_equal(free[g][0], capped[g][0]) and np.array_equal(free[g][1], capped[g][1])):
            continue                                   # (sampled: CPython's random.sample is not reproducible)
        gun, gvn, gulab, gvlab, gt = golden_canonical(case['recs'][g])
        tu, tv, tul, tvl, te = capped[g]
        assert sorted(tu) == sorted(gun.tolist()) and sorted(tv) == sorted(gvn.tolist()) and tul == gulab and tvl == gvlab
        assert te == [tuple(int(x) for x in e) for e in gt]
        compared += 1
    assert compared > 0 or name in ('synth_cap', 'douban_cap20', 'synth_h2_ratio')
