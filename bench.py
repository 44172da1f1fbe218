"""Benchmark of the IGMC hot path on MI355X: enclosing-subgraphs/sec through a full train step
(extraction -> forward -> loss(+ARR) -> backward -> [flat RCCL all-reduce] -> fused Adam), batch 50 per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): ml_1m-shaped graph, h=1, max-nodes-per-hop 100, batch 50, adj-dropout 0,
dynamic-train.  MovieLens is not available offline, so the graph is the MovieLens-shaped synthetic generator of
SURVEY.md 8(d) (6040 x 3706, 1 000 209 ratings, 90/10 split) unless raw_data/ml_1m/ratings.dat exists.
Weights are random-init (reference init).  Inputs (graph, link arrays) are resident in HBM before timing.

The ONE JSON line printed by rank 0 also carries
  roofline      : the dominant kernel = k_graph_step (forward + backward of every subgraph, one workgroup cluster
                  each; the three conv layers in both directions = 6 x the per-layer gather bytes 133*E + 132*N of
                  SURVEY.md 8(d)) -- or, for configurations it does not take, the fused forward layer kernel
                  k_rgcn_layer_fwd (1 x those bytes): algorithmic bytes per launch / its average duration measured
                  with HIP events on the launch stream in a separate instrumented pass of the same steps;
  cpu_baseline  : the oracle's restatement of the reference CPU path (scipy/python extraction + PyG-1.4.2
                  per-edge-weight formulation in torch, all host cores) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from igmc_amd import _lib, engine, parallel, preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured-achievable
BATCH = 50
CONFIGS = {
    'ml_1m': dict(dataset='ml_1m', mnph=100, adj_dropout=0.0),
    'ml_100k': dict(dataset='ml_100k', mnph=200, adj_dropout=0.2),
}


def cpu_baseline(A, tr_u, tr_v, tr_l, class_values, mnph, adj_dropout, budget_s=20.0):
    """Reference CPU path restated by the oracle (kind='port'), timed on this box's host cores."""
    import random
    from oracle import extract_ref, pyg_ref
    ncpu = os.cpu_count() or 1
    Acsc = A.tocsc()
    torch.manual_seed(1)
    random.seed(1)
    model = pyg_ref.IGMCRef(4, (32, 32, 32, 32), len(class_values), 4, adj_dropout=adj_dropout, fast=False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(tr_u))

    def make_batch(i):
        idx = perm[i * BATCH:(i + 1) * BATCH]
        graphs = [extract_ref.extract((tr_u[k], tr_v[k]), A, Acsc, 1, 1.0, mnph, class_values, tr_l[k]) for k in idx]
        return pyg_ref.Batch.from_data_list(graphs)

    def one_step(i):
        return pyg_ref.train_step(model, opt, make_batch(i), ARR=0.001)
    # the reference uses every host core; over-subscription hurts this formulation on many-core hosts, so the
    # baseline gets the best of a few thread counts (reported as `cores`)
    b0 = make_batch(0)
    best, cores = None, 1
    for th in sorted(set([min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)])):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        pyg_ref.train_step(model, opt, b0, ARR=0.001)
        el = time.perf_counter() - t0
        if best is None or el < best:
            best, cores = el, th
        if el > 8.0:
            break
    torch.set_num_threads(cores)
    first = best
    steps, t1 = 0, time.perf_counter()
    while True:
        one_step(steps + 1)
        steps += 1
        el = time.perf_counter() - t1
        if el > budget_s or el + first > budget_s * 1.5 or steps >= 50:
            break
    rate = steps * BATCH / (time.perf_counter() - t1)
    return dict(value=rate, unit='subgraphs/s', cores=cores, kind='port',
                sample='%d train steps of batch %d (extraction + PyG-1.4.2-formulation fwd/bwd + Adam), '
                       'oracle/extract_ref.py + oracle/pyg_ref.py, torch threads=%d' % (steps, BATCH, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='ml_1m', choices=sorted(CONFIGS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-steps', type=int, default=20)
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly (no hipGraph replay)')
    ap.add_argument('--no-overlap', action='store_true', help='extract batch t+1 on the same stream (no overlap)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank, world = parallel.init_from_env('nccl')
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write('warning: --gpus %d but WORLD_SIZE %d (using WORLD_SIZE)\n' % (args.gpus, world))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if os.environ.get('IGMC_LIB_PATH'):                # debug hook: time an experimental build of the library
        _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
    lib = _lib.load()                                  # fails loudly without the gfx950 library

    # ---- workload (identical on every rank)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # stdout carries the ONE JSON line only
        split = preprocessing.create_trainvaltest_split(cfg['dataset'], 1234, True, verbose=(rank == 0))
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, te_l, te_u, te_v, class_values) = split
    source = 'real' if preprocessing._load_real_movielens(cfg['dataset']) is not None else 'synthetic'
    torch.manual_seed(1)
    ds = MyDynamicDataset('data/bench', A, (tr_u, tr_v), tr_l, 1, 1.0, cfg['mnph'], None, None, class_values,
                          device=local, seed=1)
    model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(class_values), num_bases=4, regression=True,
                 adj_dropout=cfg['adj_dropout'], multiply_by=1, seed=1).to(dev)
    model.reset_parameters()
    if world > 1:
        parallel.broadcast_(model.flat_parameters(), 0)
    opt = FlatAdam(model, lr=1e-3)
    loss = None
    n = len(ds)
    gen = torch.Generator()
    gen.manual_seed(1234)
    perm_all = torch.randperm(n, generator=gen)
    perm = parallel.shard_positions(perm_all, rank, world, pad=True)
    steps_avail = len(perm) // BATCH
    st = torch.cuda.current_stream().cuda_stream
    # the product's training path: one optimisation step = hipGraph replay of
    # { forward/backward/finalize (batch t)  ||  extract batch t+1 } -> [flat all-reduce] -> Adam+loss+tick
    sg = StepGraph(model, opt, ds, BATCH, 0.001, use_graph=not args.no_graph, overlap=not args.no_overlap)
    state = dict(i=0, epoch=0)

    def step():
        if state['i'] % steps_avail == 0:
            state['epoch'] += 1
            sg.begin_epoch(perm, state['epoch'])
        state['i'] += 1
        sg.step()
        return sg

    def run(nsteps):                    # exactly `nsteps` optimisation steps, epoch after epoch
        while nsteps > 0:
            if state['i'] % steps_avail == 0:
                state['epoch'] += 1
                sg.begin_epoch(perm, state['epoch'])
            chunk = min(nsteps, steps_avail - state['i'] % steps_avail)
            sg.steps(chunk)
            state['i'] += chunk
            nsteps -= chunk

    run(args.warmup)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    value = args.steps * BATCH * world / dt
    sg.check()                                          # no device-side wait timed out during the timed steps
    final_loss = float(sg.loss[0].item())

    # ---- roofline leg: instrumented pass (HIP events around every kernel on the launch stream)
    roofline = None
    kernels = {}
    if args.profile_steps > 0:
        # EVERY rank runs the instrumented steps (under data parallelism each of them holds a gradient all-reduce,
        # a collective all ranks must enter); rank 0 reports its own kernels
        engine.profile_enable(lib, True)
        Ns, Es = [], []
        sg.use_graph, sg.graphs = False, [None, None]   # instrumented pass launches eagerly
        for _ in range(args.profile_steps):
            step()
            info = sg.arena.info(st)
            Ns.append(info.num_nodes)
            Es.append(info.num_edges)
        torch.cuda.synchronize()
        rows = engine.profile_fetch(lib, 64)
        engine.profile_enable(lib, False)
        N, E = float(np.mean(Ns)), float(np.mean(Es))
        kernels = {name: dict(us=ms / calls * 1e3, calls_per_step=calls / args.profile_steps) for name, ms, calls in rows}
        tot = sum(ms for _, ms, _ in rows)
        dom = 'k_graph_step' if 'k_graph_step' in kernels else (
            'k_rgcn_layer_fwd' if 'k_rgcn_layer_fwd' in kernels else 'k_rgcn_gather_fwd')
        if dom in kernels and args.profile_steps > 0:
            algo_bytes = 133.0 * E + 132.0 * N                  # one layer, one direction
            if dom == 'k_graph_step':                           # forward + backward of the 3 conv layers in one launch
                algo_bytes *= 6.0
            dur_s = kernels[dom]['us'] * 1e-6
            achieved = algo_bytes / dur_s / 1e9
            # HBM-side bytes per launch of this kernel: PMC counters cannot be read from inside this process, they come
            # from the committed rocprofv3 --pmc passes of the same command (tools/profile_round.sh -> pmc_traffic.py)
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
            if args.config == 'ml_1m' and os.path.exists(tpath):
                trec = json.load(open(tpath))
                if trec.get('kernel') == dom:
                    traffic = trec.get('traffic_bytes')
            roofline = dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=achieved / HBM_PEAK_GBS, traffic=traffic, avg_us=kernels[dom]['us'],
                            algorithmic_bytes=algo_bytes, nodes=N, edges=E,
                            share_of_kernel_time=kernels[dom]['us'] * kernels[dom]['calls_per_step'] * args.profile_steps
                            / (tot * 1e3) if tot > 0 else None)
    if world > 1:
        parallel.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(A, tr_u, tr_v, tr_l, class_values, cfg['mnph'], cfg['adj_dropout'])
        except MemoryError:
            cpu = None
    if rank == 0:
        rec = {
            'metric': 'enclosing-subgraphs/sec (train step, batch=50)',
            'value': value, 'unit': 'subgraphs/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': '%s %s-shaped rating graph (%d x %d, %d train links), random-init weights' % (
                source, cfg['dataset'], A.shape[0], A.shape[1], n),
            'config': {'workload': '%s, hop 1, max-nodes-per-hop %d, batch %d per GPU, adj-dropout %g, dynamic-train, '
                                   'ARR 0.001, Adam' % (cfg['dataset'], cfg['mnph'], BATCH, cfg['adj_dropout']),
                       'parallelism': 'dp%d' % world, 'global_batch': BATCH * world},
            'roofline': roofline, 'cpu_baseline': cpu, 'final_loss': final_loss,
            'kernels_us': {k: round(v['us'], 2) for k, v in kernels.items()},
        }
        print(json.dumps(rec))
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
