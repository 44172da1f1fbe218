"""Run-to-run and structure-to-structure bit comparisons of the training step graph on the GPU (round 3).

For each configuration (ml_1m shape with / without edge dropout, douban with edge dropout; lean arenas, dropout drawn on the
dense blocks inside the graph) a reference trajectory of two epochs x 24 steps (groups of 8: graph launches of 16 steps)
is compared bit for bit -- parameters, both Adam moments, both epoch totals -- with N further trajectories that alternate
between the launch structures (groups of 8 again, groups of 4, groups of 2, eager one-stream).  Every trajectory ends in
StepGraph.check() (stamp mismatches / timed-out waits raise).  Prints one line per comparison and a summary; exit code 1
if any comparison differs.

    python tools/exp_repro.py [N]        (default 20 comparisons per configuration)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd.hostcpu import limit_host_threads  # noqa: E402
limit_host_threads()
import numpy as np  # noqa: E402
import torch  # noqa: E402
from igmc_amd import preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402


def trajectory(ds, drop, perm, **kw):
    torch.manual_seed(3)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=drop,
                 seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001, **kw)
    totals = []
    for ep in (1, 2):
        t, _ = sg.run_epoch(perm, ep)
        totals.append(float(t.item()))
    torch.cuda.synchronize()
    return (model.flat_parameters().detach().cpu().clone(), opt.exp_avg.detach().cpu().clone(),
            opt.exp_avg_sq.detach().cpu().clone(), totals), sg.graph is not None


def describe(a, b):
    out = []
    for name, x, y in zip(('params', 'exp_avg', 'exp_avg_sq'), a[:3], b[:3]):
        if not torch.equal(x, y):
            out.append('%s: %d of %d differ, max |d| %.3e' % (name, int((x != y).sum()), x.numel(), float((x - y).abs().max())))
    if a[3] != b[3]:
        out.append('epoch totals %r vs %r' % (a[3], b[3]))
    return '; '.join(out)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
    pick = np.random.default_rng(11).permutation(len(tr_u))[:1200]
    ml = MyDynamicDataset('data/t/exp_repro_ml', A, (tr_u[pick], tr_v[pick]), np.asarray(tr_l)[pick], 1, 1.0, 100, None,
                          None, cv, device=0, seed=1)
    (_, _, Ad, dl, du, dv, _, _, _, _, _, _, cvd) = preprocessing.load_data_monti('douban', testing=True)
    db = MyDynamicDataset('data/t/exp_repro_db', Ad, (du[:1200], dv[:1200]), dl[:1200], 1, 1.0, 10000, None, None, cvd,
                          device=0, seed=1)
    structures = [('groups of 8', dict(group=8)), ('groups of 4', dict(group=4)), ('groups of 2', dict(group=2)),
                  ('eager, one stream', dict(use_graph=False, overlap=False, group=8)),
                  ('eager, two streams', dict(use_graph=False, overlap=True, group=8))]
    bad = 0
    for cname, ds, drop in (('ml_1m shape, cap 100, edge dropout 0.2', ml, 0.2), ('ml_1m shape, cap 100, no edge dropout', ml, 0.0),
                            ('douban, uncapped, edge dropout 0.2', db, 0.2)):
        perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(5))
        ref, captured = trajectory(ds, drop, perm, group=8)
        assert captured
        same = 0
        for i in range(N):
            sname, kw = structures[i % len(structures)]
            try:
                got, _ = trajectory(ds, drop, perm, **kw)
                diff = describe(ref, got)
            except RuntimeError as e:
                diff = 'RAISED: %s' % str(e).splitlines()[0]
            print('%-42s #%02d %-20s %s' % (cname, i, sname, 'identical' if not diff else 'DIFFERENT: ' + diff), flush=True)
            same += not diff
        print('== %s: %d of %d comparisons identical' % (cname, same, N), flush=True)
        bad += N - same
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
