"""Experiment (GPU): the step chain and the extraction chain of one group of M steps as TWO LINEAR hipGraphs launched one after the
other from the host (fork / join as stream waits) against the product's ONE two-branch graph: host time of the launches and the
wall-clock start of the first extraction launch relative to the first step (clock marks captured into the graphs)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from igmc_amd import preprocessing
from igmc_amd.models import IGMC
from igmc_amd.stepgraph import StepGraph
from igmc_amd.train_eval import FlatAdam
from igmc_amd.util_functions import MyDynamicDataset
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mk = C.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libclock_mark.so'))
mk.clock_mark.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
split = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
ds = MyDynamicDataset('data/x', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, cv, device=0, seed=1)
model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=5, num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
model.reset_parameters()
opt = FlatAdam(model, lr=1e-3)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))
sg = StepGraph(model, opt, ds, 50, 0.001, group=M)
buf = torch.zeros(64, dtype=torch.int64, device='cuda')
sg.begin_epoch(perm, 1)
sg.steps(1)
sg.prepare(steps_hint=M)            # the product's single-group graphs (two-branch), primed
assert sg.single_launch and sg.graph1[0] is not None


def mark(slot):
    mk.clock_mark(C.c_void_p(buf.data_ptr()), slot, C.c_void_p(torch.cuda.current_stream().cuda_stream))


def capture_split(q):
    cur = [sg._arena(q, i) for i in range(M)]
    plan = sg._extract_plan(1 - q, M)
    per = max(1, M // len(plan))
    gm, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gm):
        mark(0 + 4 * q)
        for i in range(M):
            if i > 0:
                sg._hint_unchanged()
            sg._enqueue_step(cur[i], sg.B)
        mark(1 + 4 * q)
    with torch.cuda.graph(gs):
        for j, fn in enumerate(plan):
            st = torch.cuda.current_stream().cuda_stream
            sg.lib.call('igmc_ctrl_gate', C.c_void_p(sg.ctrl.data_ptr()), q, j * per, 10.0, 1 if j == 0 else 0, 2000.0, C.c_void_p(st))
            if j == 0:
                mark(2 + 4 * q)
            fn()
        mark(3 + 4 * q)
    return gm, gs


G = [capture_split(0), capture_split(1)]
main, side = torch.cuda.current_stream(), sg.side


def launch_split(q):
    gm, gs = G[q]
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gs.replay()
    gm.replay()
    main.wait_stream(side)


def run(label, fn, reps=6):
    # always start at parity 0 with its arenas filled
    out = []
    for r in range(reps):
        sg._regroup()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        fn(0)
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        sg.k += M; sg.steps_done += M; sg._count(M)
        sg.gq, sg.gk, sg.avail = 1, 0, M
        t = buf[:4].cpu().numpy().astype('int64')
        out.append((e0.elapsed_time(e1) * 1e3 / M, (t1 - t0) * 1e6, (t[2] - t[0]) / 100.0, (t[3] - t[0]) / 100.0, (t[1] - t[0]) / 100.0))
    print(label)
    for o in out:
        print('   %.2f us/step by events | host %.0f us | first extraction launch +%.0f us, extraction chain end +%.0f us, last step end +%.0f us' % o)
    sg.check()


run('two linear graphs (side launched first), M = %d' % M, launch_split)


def launch_product(q):
    sg.graph1[q].replay()


buf.zero_()
run('the product: one two-branch graph, M = %d  (no marks inside: offsets are stale)' % M, launch_product)
