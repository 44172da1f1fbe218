"""Free-running training: the engine's RNG path (hash sampler, hash dropout) against the reference's (``random.sample`` +
torch RNG; restated by the oracle) -- agreement of the final test RMSE IN DISTRIBUTION on a small capped configuration, ten
seeds a side (``free_run_parity.py``).  Both sides are deterministic given their seeds, so the
outcome is a property of the code, not of the day."""
import pytest

import free_run_parity as F

pytestmark = pytest.mark.gpu


def test_free_running_test_rmse_agrees_with_the_reference_rng_path_in_distribution(monkeypatch):
    # (eager launches: the trajectories are bit-identical to the hipGraph replays -- test_step_graph_is_bit_reproducible_and_
    #  structure_independent -- and twenty graph captures less keep this file's run short)
    monkeypatch.setenv('IGMC_NO_GRAPH', '1')
    monkeypatch.setenv('IGMC_NO_EVAL_GRAPH', '1')
    D = F.make_data()
    oracle = F.oracle_runs()
    engine = [F.engine_run(D, s) for s in F.SEEDS]
    diff, se, lines = F.compare(engine, oracle)
    print('\n'.join(lines))
    # both learn (the ratings' own spread is 1.10) ...
    assert max(engine) < 0.85 and max(oracle) < 0.85, lines
    # ... and the means agree within one standard deviation of a run -- and within 3 standard errors
    import numpy as np
    sd_run = float(np.sqrt((np.var(engine, ddof=1) + np.var(oracle, ddof=1)) / 2))
    assert diff <= max(sd_run, 0.004) and diff <= 3.0 * se + 0.002, lines
