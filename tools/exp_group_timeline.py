"""Experiment (GPU): the position in time of every step and every extraction launch INSIDE one replay of the step graph,
without a profiler attached: one-thread kernels that write the wall clock (tools/ubench/clock_mark.hip) are captured in front
of and behind every step on the step chain and every extraction launch on the extraction chain.

    python tools/exp_group_timeline.py [--group 10] [--side-first]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from igmc_amd import preprocessing  # noqa: E402
from igmc_amd.models import IGMC  # noqa: E402
from igmc_amd.stepgraph import StepGraph  # noqa: E402
from igmc_amd.train_eval import FlatAdam  # noqa: E402
from igmc_amd.util_functions import MyDynamicDataset  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--group', type=int, default=10)
ap.add_argument('--single', type=int, default=0, help='a run of that many steps (<= 32) as ONE single-group launch (round 6)')
ap.add_argument('--delay-us', type=float, default=0.0, help='a spin kernel of that length in front of every extraction launch (behind its gate)')
ap.add_argument('--config', default='ml_1m', help="dataset (preprocessing.create_trainvaltest_split): ml_1m, ml_10m_lite, ...")
args = ap.parse_args()
if os.environ.get('IGMC_LIB_PATH'):                # debug hook: an experimental build of the library
    from igmc_amd import _lib
    _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
M = args.group
mk = C.CDLL(os.path.join(ROOT, 'tools', 'ubench', 'libclock_mark.so'))
mk.clock_mark.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
split = preprocessing.create_trainvaltest_split(args.config, 1234, True, verbose=False)
(_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = split
torch.cuda.set_device(0)
ds = MyDynamicDataset('data/x', A, (tr_u, tr_v), tr_l, 1, 1.0, 100, None, None, cv, device=0, seed=1)
model = IGMC(ds, latent_dim=[32, 32, 32, 32], num_relations=len(cv), num_bases=4, regression=True, adj_dropout=0.0, seed=1).to('cuda')
model.reset_parameters()
opt = FlatAdam(model, lr=1e-3)
perm = torch.randperm(len(ds), generator=torch.Generator().manual_seed(1))
sg = StepGraph(model, opt, ds, 50, 0.001)
buf = torch.zeros(4096, dtype=torch.int64, device='cuda')
labels = []
rec = [False]


def mark(label):
    if rec[0]:
        mk.clock_mark(C.c_void_p(buf.data_ptr()), len(labels), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        labels.append(label)


step_no, side_no = [0], [0]
orig_step, orig_side = sg._enqueue_step, sg._side


def enqueue_step(arena, B):
    mark('step %2d start' % step_no[0])
    orig_step(arena, B)
    mark('step %2d end' % step_no[0])
    step_no[0] += 1


def side(fn):
    def wrapped():
        if args.delay_us > 0 and rec[0]:
            torch.cuda._sleep(int(args.delay_us * 2350))
        mark('        extraction %2d start' % side_no[0])
        fn()
        mark('        extraction %2d end' % side_no[0])
        side_no[0] += 1
    orig_side(wrapped)


sg._enqueue_step, sg._side = enqueue_step, side
sg.begin_epoch(perm, 1)
sg.step()
rec[0] = True
if args.single:
    M = args.single
    sg.prepare(steps_hint=M)
    rec[0] = False
    n_marks = len(labels) // 2          # (both parities' graphs were captured: the marks of the first one)
    N = M
    sg.steps(2 * M)                     # parity 0, parity 1: back at the first graph
else:
    sg.prepare(group=M)
    rec[0] = False
    n_marks = len(labels)
    N = 2 * M
    for _ in range(4):
        sg.steps(N)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
sg.steps(N)
ev1.record()
torch.cuda.synchronize()
t = buf[:n_marks].cpu().numpy().astype('int64')
print('# IGMC_EXTRACT_PACED=%s delay %.0f us' % (os.environ.get('IGMC_EXTRACT_PACED', 'default'), args.delay_us))
print('# M = %d, %d marks; the replay between two synchronizes: %.2f us/step by HIP events (with the marks in it)'
      % (M, n_marks, ev0.elapsed_time(ev1) * 1e3 / N))
order = sorted(range(n_marks), key=lambda i: t[i])
t0 = t[order[0]]
for i in order:
    print('%9.2f us  %s' % ((t[i] - t0) / 100.0, labels[i]))
sg.check()
