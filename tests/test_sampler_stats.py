"""The per-hop sampler's DISTRIBUTION (reference ``util_functions.py:222-229``: ``random.sample`` = a uniform k-subset per
link, hop, side and epoch) -- see ``sampler_stats.py`` for the statistics.  4 200 draws per link (100 link positions x 2
dataset seeds x 21 epochs) of three capped links, n = 150 / 400 / 2 000 candidates a side, k = 100.

CPU: the draws come from ``oracle/extract_cpu.c`` (the twin the HIP kernels are held to bit for bit), the emulated HIP kernels
reproduce a slice of them; GPU: the real kernels produce all of them, bit-equal to the twin.  The sampler is a deterministic
hash, so these tests cannot flake: they pass or fail for good."""
import numpy as np
import pytest

import parity_checks as PC
import sampler_stats as S


def _bounds(worst):
    assert worst['p_min'] > 1e-3, worst               # singles: chi-square of the inclusion counts vs k / n
    assert worst['z_pairs'] < 5.0, worst              # pairs: co-inclusion vs k (k - 1) / (n (n - 1))
    assert worst['z_sides'] < 5.0, worst              # the two sides of a draw are independent
    assert worst['z_epochs'] < 5.0, worst             # consecutive epochs of a link are independent
    assert worst['z_rank'] < 5.0, worst               # no id bias


def test_sampler_draws_are_distributed_like_uniform_k_subsets():
    draws = S.draws_from_twin()
    lines, worst = S.report(draws)
    print('\n'.join(lines))
    _bounds(worst)
    # two dataset seeds, two epochs, two link positions of the same pair: all different draws
    a = draws[(1, 1)][0]
    assert not np.array_equal(a[0], a[3]) and not np.array_equal(a, draws[(2, 1)][0]) and not np.array_equal(a, draws[(1, 2)][0])


def test_the_statistics_reject_a_lowest_id_sampler():
    """Negative control: the k lowest candidate ids every time -- every bound is violated by orders of magnitude."""
    _, worst = S.report(S.lowest_id_draws(), reps=3)
    assert worst['p_min'] < 1e-12 and min(worst['z_pairs'], worst['z_sides'], worst['z_epochs'], worst['z_rank']) > 100.0, worst


def test_emulated_kernels_draw_what_the_twin_draws():
    be = PC.EmuBackend()
    seeds, epochs = (1,), (1, 2)
    eng, twin = S.draws_from_engine(be, seeds, epochs), S.draws_from_twin(seeds, epochs)
    for key in twin:
        assert np.array_equal(eng[key][0], twin[key][0]) and np.array_equal(eng[key][1], twin[key][1]), key


def test_check_sampled_rejects_an_id_biased_sampler():
    """``parity_checks.check_sampled`` -- what the free-running capped cases are held to -- must not pass a sampler that takes the
    lowest candidate ids: the engine REPLAYS such node lists here (so sizes, membership and induced edges are all right)."""
    from helpers import load_extract_golden
    case = dict(load_extract_golden()['synth_cap'])
    A, Acsc = case['A'].tocsr(), case['A'].tocsc()
    recs = []
    for (i, j), rec in zip(case['links'], case['recs']):
        cu = sorted(set(Acsc.indices[Acsc.indptr[j]:Acsc.indptr[j + 1]].tolist()) - {int(i)})[:len(rec['u_nodes']) - 1]
        cv = sorted(set(A.indices[A.indptr[i]:A.indptr[i + 1]].tolist()) - {int(j)})[:len(rec['v_nodes']) - 1]
        un, vn = np.array([i] + cu), np.array([j] + cv)
        labels = np.array([0] + [2] * len(cu) + [1] + [3] * len(cv))
        recs.append(dict(rec, u_nodes=un, v_nodes=vn, labels=labels))
    case['recs'] = recs
    be = PC.EmuBackend()
    _, _, d = PC.extract_case(be, case, replay=True)
    with pytest.raises(AssertionError, match='not uniform over the candidate ids'):
        PC.check_sampled(d, case)
    _, _, d = PC.extract_case(be, dict(case, recs=load_extract_golden()['synth_cap']['recs']), replay=False, seed=5, epoch=1)
    PC.check_sampled(d, load_extract_golden()['synth_cap'])          # ... and passes the engine's own draws


@pytest.mark.gpu
def test_gpu_kernels_draw_what_the_twin_draws_and_the_draws_are_uniform():
    be = PC.GpuBackend()
    eng, twin = S.draws_from_engine(be), S.draws_from_twin()
    for key in twin:
        assert np.array_equal(eng[key][0], twin[key][0]) and np.array_equal(eng[key][1], twin[key][1]), key
    lines, worst = S.report(eng, reps=6)
    print('\n'.join(lines))
    _bounds(worst)
