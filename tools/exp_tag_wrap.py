"""Long runs in one process per path (70 k+ training launches): the exchange flags carry the launch sequence number (32 bits
since round 5: no tag wrap, no periodic clear of the exchange regions -- rounds 3 / 4 had 16-bit tags in the data words);
bounded polls and the arena stamps are checked by StepGraph.check() after every epoch, with evaluation batches between the
epochs (they advance the sequence number by one: both parities).
   python tools/exp_tag_wrap.py [ml_1m|ml_100k|flixster ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from igmc_amd import preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402


def run(cfgname, want_steps=70000):
    cfg = bench.CONFIGS[cfgname]
    if cfg['dataset'] in ('douban', 'flixster', 'yahoo_music'):
        split = preprocessing.load_data_monti(cfg['dataset'], testing=True)
    else:
        rmap = {float(i): i / 2.0 for i in range(1, 11)} if cfg['dataset'] == 'ml_10m_lite' else None
        split = preprocessing.create_trainvaltest_split(cfg['dataset'], 1234, True, rating_map=rmap, verbose=False)
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, class_values) = split
    ds = MyDynamicDataset('data/wrap_%s' % cfgname, A, (tr_u, tr_v), tr_l, 1, 1.0, cfg['mnph'], None, None, class_values,
                          device=0, seed=1)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(class_values), num_bases=4, regression=True,
                 adj_dropout=cfg['adj_dropout'], multiply_by=1, seed=1).to('cuda')
    model.reset_parameters()
    opt = FlatAdam(model, lr=1e-3)
    sg = StepGraph(model, opt, ds, 50, 0.001)
    steps, ep, t0, losses = 0, 0, time.time(), []
    gen = torch.Generator().manual_seed(3)
    while steps < want_steps:
        ep += 1
        perm = torch.randperm(len(ds), generator=gen)
        total, n = sg.run_epoch(perm, ep)           # (check() inside: bounded polls, arena stamps)
        losses.append(float(total.item()) / n)
        steps += (n + 49) // 50
    torch.cuda.synchronize()
    ok = all(np.isfinite(losses)) and losses[-1] < losses[0]
    print('%s: %d steps in %d epochs, %.1f s; mean loss of epoch 1 / last: %.4f / %.4f; pacing gates %s; %s'
          % (cfgname, steps, ep, time.time() - t0, losses[0], losses[-1],
             'fell back to edges' if sg.pacing_fallback else 'never gave up often enough to fall back', 'OK' if ok else 'SUSPECT'))
    return ok


if __name__ == '__main__':
    names = sys.argv[1:] or ['ml_1m', 'ml_100k', 'flixster']
    sys.exit(0 if all([run(n) for n in names]) else 1)
