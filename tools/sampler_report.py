"""profiles/r06_sampler_stats.txt: the per-hop sampler's distribution statistics on the REAL kernels (tests/sampler_stats.py)
and the free-running statistical parity run (tests/free_run_parity.py).  python tools/sampler_report.py > out.txt (on a GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402
import parity_checks as PC  # noqa: E402
import sampler_stats as S  # noqa: E402
import free_run_parity as F  # noqa: E402

if __name__ == '__main__':
    print('# per-hop sampler (k smallest igmc_sample_key) vs uniform k-subsets (reference util_functions.py:222-229, random.sample)')
    print('# draws: the HIP kernels on this GPU, %d link positions x %d seeds x %d epochs per link; null of the calibrated statistics: numpy uniform k-subsets'
          % (S.POS_PER_LINK, len(S.SEEDS), len(S.EPOCHS)))
    be = PC.GpuBackend()
    t0 = time.time()
    eng, twin = S.draws_from_engine(be), S.draws_from_twin()
    same = all(np.array_equal(eng[k][0], twin[k][0]) and np.array_equal(eng[k][1], twin[k][1]) for k in twin)
    print('# GPU draws == oracle/extract_cpu.c draws, bit for bit: %s (%.1f s)' % (same, time.time() - t0))
    lines, worst = S.report(eng)
    print('\n'.join(lines))
    print('worst: singles p_min %.3f (bound > 1e-3); |z| pairs %.2f, sides %.2f, epochs %.2f, id-rank %.2f (bound < 5)'
          % (worst['p_min'], worst['z_pairs'], worst['z_sides'], worst['z_epochs'], worst['z_rank']))
    _, bad = S.report(S.lowest_id_draws(), reps=3)
    print('negative control (k lowest ids every time): p_min %.1e, |z| pairs %.0f, sides %.0f, epochs %.0f, id-rank %.0f'
          % (bad['p_min'], bad['z_pairs'], bad['z_sides'], bad['z_epochs'], bad['z_rank']))
    print()
    print('# free-running training, final test RMSE after %d epochs: %d x %d ratings matrix, max-nodes-per-hop %d, adj-dropout %g, batch %d, %d train / %d test links, seeds %s'
          % (F.EPOCHS, F.N_USERS, F.N_ITEMS, F.CAP, F.ADJ_DROPOUT, F.BATCH, F.N_TRAIN, F.N_TEST, list(F.SEEDS)))
    D = F.make_data()
    t0 = time.time()
    oracle = F.oracle_runs()
    t1 = time.time()
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        engine = [F.engine_run(D, s, tag='rep') for s in F.SEEDS]
    t2 = time.time()
    _, _, lines = F.compare(engine, oracle)
    print('\n'.join(lines))
    print('# wall: oracle %.1f s (worker processes), engine %.1f s' % (t1 - t0, t2 - t1))
