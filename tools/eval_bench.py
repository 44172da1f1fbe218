"""Evaluation throughput (train_eval.eval_rmse through EvalGraph) at the headline shape: python tools/eval_bench.py [--links N] [--config ml_1m]
[--dynamic].  Prints subgraphs/s of the second and third evaluation (the first captures the graph)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd.hostcpu import limit_host_threads  # noqa: E402
limit_host_threads()
import torch  # noqa: E402
from igmc_amd import preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.train_eval import DataLoader, eval_rmse  # noqa: E402
from igmc_amd.util_functions import MyDataset, MyDynamicDataset  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--links', type=int, default=20000)
    ap.add_argument('--config', default='ml_1m')
    ap.add_argument('--mnph', type=int, default=100)
    ap.add_argument('--dynamic', action='store_true')
    ap.add_argument('--reps', type=int, default=3)
    a = ap.parse_args()
    if os.environ.get('IGMC_LIB_PATH'):                # debug hook: an experimental build of the library
        from igmc_amd import _lib
        _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        split = preprocessing.create_trainvaltest_split(a.config, 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, te_l, te_u, te_v, cv) = split
    m = min(a.links, len(te_u))
    cls = MyDynamicDataset if a.dynamic else MyDataset
    te = cls('data/evalbench', A, (te_u[:m], te_v[:m]), te_l[:m], 1, 1.0, a.mnph, None, None, cv, seed=1)
    torch.manual_seed(1)
    model = IGMC(te, latent_dim=[32, 32, 32, 32], num_relations=len(cv), num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
    model.reset_parameters()
    model.eval()
    tl = DataLoader(te, 50, shuffle=False)
    vals = []
    for r in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v = eval_rmse(model, tl, 'cuda')
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        vals.append(v)
        print('eval %d: %d links in %.3f ms = %.0f subgraphs/s (%.2f us/step), rmse %.6f' % (r, m, dt * 1e3, m / dt, dt / (m / 50) * 1e6, v))
    eg = getattr(tl, '_evalgraph', None)
    if eg is not None:
        print('group M = %d, graph %s, lean %s' % (eg.M, eg.graph is not None, eg._lean))
