"""CPU-only tests of the host side: the C-ABI library loads and exports every declared symbol (no compute
calls), loud failure without the library, data-parallel host logic under gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT


def test_c_abi_exports_every_declared_symbol():
    from igmc_amd import _lib, build
    build.build_hip()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, 'include', 'igmc_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(igmc_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib.cdll, name), 'libigmc_hip.so does not export %s' % name
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
    assert lib.igmc_version() >= 100


def test_missing_library_fails_loudly(tmp_path):
    from igmc_amd import _lib
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load(str(tmp_path / 'nope.so'))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under igmc_amd/, Main.py may import it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'igmc_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M):
                    bad.append(f)
    src = open(os.path.join(ROOT, 'Main.py')).read()
    assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M)
    assert not bad, bad


def test_bench_never_runs_fewer_ranks_than_asked_for():
    """``python bench.py --gpus N`` without a launcher spawns its N ranks itself; with fewer than N visible devices it refuses
    (rc != 0, nothing on stdout) instead of measuring one GPU under an N-GPU label; a launcher that started another number
    of ranks than ``--gpus`` is refused the same way."""
    import subprocess
    bench = os.path.join(ROOT, 'bench.py')
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'IGMC_LOCAL_DEVICE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '4', '--warmup', '1'], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == b'' and b'refusing' in r.stderr, (r.returncode, r.stderr[-500:])
    r = subprocess.run([sys.executable, bench, '--gpus', '4', '--steps', '4', '--warmup', '1'], env=dict(env, WORLD_SIZE='2', RANK='0'),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == b'' and b'WORLD_SIZE=2' in r.stderr, (r.returncode, r.stderr[-500:])


def test_shard_positions():
    import torch
    from igmc_amd import parallel
    perm = torch.randperm(103)
    shards = [parallel.shard_positions(perm, r, 4, pad=True) for r in range(4)]
    assert all(len(s) == 26 for s in shards)                       # equal step counts on every rank
    assert set(torch.cat(shards).tolist()) == set(range(103))
    ev = [parallel.shard_positions(perm, r, 4, pad=False) for r in range(4)]
    assert sorted(torch.cat(ev).tolist()) == list(range(103))      # eval: nothing counted twice
    assert parallel.shard_positions(perm, 0, 1) is perm


_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from igmc_amd import parallel
rank, world = parallel.init_from_env('gloo')
assert world == 2 and parallel.world_size() == 2 and parallel.rank() == rank
# flat-gradient all-reduce: global-batch mean = sum over ranks of (local sum / (B*G)); ARR term scaled 1/G
torch.manual_seed(0)
full = torch.randn(4, 1000)                       # per-sample gradients of a global batch of 4
arr = torch.randn(1000)
local = full[rank * 2:(rank + 1) * 2].sum(0) / 4 + arr / world
parallel.all_reduce_sum_(local)
ref = full.mean(0) + arr
assert torch.allclose(local, ref, atol=1e-6), (local - ref).abs().max()
# identical replicas after broadcast
p = torch.full((10,), float(rank))
parallel.broadcast_(p, 0)
assert p.sum().item() == 0.0
# eval reduction of (sse, count)
acc = torch.tensor([float(rank + 1), 10.0], dtype=torch.float64)
parallel.all_reduce_sum_(acc)
assert acc.tolist() == [3.0, 20.0]
# equal step counts with padded shards
perm = torch.arange(101)
mine = parallel.shard_positions(perm, rank, world, pad=True)
n = torch.tensor([float(len(mine))]); parallel.all_reduce_sum_(n)
assert n.item() == 2 * 51
parallel.barrier()
print('rank', rank, 'ok')
'''


def test_data_parallel_host_logic_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_WORKER % ROOT)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29611')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok' % r in o


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch
from igmc_amd import parallel, engine
import parity_checks as PC
from helpers import load_extract_golden
rank, world = parallel.init_from_env('gloo')
be = PC.EmuBackend()                      # kernel logic on the CPU emulation (test infrastructure)
case = dict(load_extract_golden()['synth_cap'])
B = 4                                     # per-rank batch; global batch = 8 links
def part(lo, hi):
    c = dict(case)
    c['recs'], c['links'], c['link_labels'] = case['recs'][lo:hi], case['links'][lo:hi], case['link_labels'][lo:hi]
    return c
rng = np.random.default_rng(0)
gmask = (rng.random((B * world, 128)) < 0.5).astype(np.uint8)
def grads(c, mask, grad_scale, arr_scale, tag):
    g, b, d = PC.extract_case(be, c, replay=True)
    ws = engine.ModelWorkspace(be.lib, be.device, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    ref = PC.make_ref_model(4, 5, seed=4)
    P = PC.flatten_params(ws, ref)
    G, out = np.zeros_like(P), np.zeros(d['B'], np.float32)
    lm = np.ascontiguousarray(mask.reshape(-1))
    ws.loss_grad(P.ctypes.data, b, out.ctypes.data, G.ctypes.data, None, lin_mask=lm.ctypes.data, ARR=0.001,
                 grad_scale=grad_scale, arr_scale=arr_scale)
    return ws, P, G
# this rank's shard: mean over the GLOBAL batch = sum over ranks of (local sum / (B * G)); ARR gradient scaled 1/G
ws, P, G = grads(part(rank * B, (rank + 1) * B), gmask[rank * B:(rank + 1) * B], 1.0 / (B * world), 1.0 / world, 'dp')
t = torch.from_numpy(G)
parallel.all_reduce_sum_(t)               # ONE flat all-reduce (reference has no distributed code; DESIGN.md section 5)
M1, M2 = np.zeros_like(P), np.zeros_like(P)
ws.adam_step(P.ctypes.data, G.ctypes.data, M1.ctypes.data, M2.ctypes.data, 1, 1e-3)
# single-process run on the whole global batch
ws1, P1, G1 = grads(part(0, B * world), gmask, 1.0 / (B * world), 1.0, 'single')
scale = np.abs(G1).max()
assert np.abs(G - G1).max() < 2e-5 * scale, np.abs(G - G1).max() / scale
M1, M2 = np.zeros_like(P1), np.zeros_like(P1)
ws1.adam_step(P1.ctypes.data, G1.ctypes.data, M1.ctypes.data, M2.ctypes.data, 1, 1e-3)
# Adam's first step is lr * sign(g) (up to eps): compare where the gradient is not at rounding level
big = np.abs(G1) > 1e-3 * scale
assert np.allclose(P[big], P1[big], rtol=0, atol=2e-6)
# replicas stay identical: every rank holds the same parameters after the step
chk = torch.from_numpy(P.copy()); parallel.broadcast_(chk, 0)
assert np.array_equal(chk.numpy(), P)
parallel.barrier()
print('rank', rank, 'dp ok')
'''


def test_data_parallel_step_equals_global_batch_step_gloo(tmp_path):
    """Two ranks (gloo), each running the kernels' logic on ITS half of a global batch with the data-parallel scale
    arguments, one flat all-reduce, Adam: the result equals one process stepping on the whole global batch."""
    from helpers import emu_lib
    emu_lib()                                   # build once, not concurrently in the two workers
    script = tmp_path / 'dp.py'
    script.write_text(_DP_WORKER % (ROOT, ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29613')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d dp ok' % r in o


_DP_RAGGED_WORKER = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch
from igmc_amd import parallel, engine
import parity_checks as PC
from helpers import load_extract_golden
rank, world = parallel.init_from_env('gloo')
be = PC.EmuBackend()
case = dict(load_extract_golden()['synth_cap'])
n, B = 13, 4                              # 13 links over 2 ranks, per-rank batch 4: shards of 7 = one full batch + 3
perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
mine = parallel.shard_positions(perm, rank, world, pad=True)
assert len(mine) == 7                     # padded: equal step counts on both ranks
last = [int(x) for x in mine[B:]]         # the ragged last batch of THIS rank (3 links)
every = [int(x) for r in range(world) for x in parallel.shard_positions(perm, r, world, pad=True)[B:]]
assert len(last) == 3 and len(every) == 6
def pick(idx):
    c = dict(case)
    c['recs'] = [case['recs'][i] for i in idx]
    c['links'], c['link_labels'] = case['links'][idx], case['link_labels'][idx]
    return c
rng = np.random.default_rng(1)
gmask = {i: (rng.random(128) < 0.5).astype(np.uint8) for i in range(n)}
def grads(idx, grad_scale, arr_scale):
    c = pick(idx)
    g, b, d = PC.extract_case(be, c, replay=True)
    ws = engine.ModelWorkspace(be.lib, be.device, 5, 4, 4, 0, b.node_capacity, b.edge_capacity, b.max_graphs)
    ref = PC.make_ref_model(4, 5, seed=4)
    P = PC.flatten_params(ws, ref)
    G, out = np.zeros_like(P), np.zeros(d['B'], np.float32)
    lm = np.ascontiguousarray(np.stack([gmask[i] for i in idx]).reshape(-1))
    ws.loss_grad(P.ctypes.data, b, out.ctypes.data, G.ctypes.data, None, lin_mask=lm.ctypes.data, ARR=0.001,
                 grad_scale=grad_scale, arr_scale=arr_scale)
    return G
# what StepGraph passes for a ragged batch of B' links per rank: grad_scale = 1 / (B' * world), ARR scaled 1 / world
G = grads(last, 1.0 / (len(last) * world), 1.0 / world)
t = torch.from_numpy(G)
parallel.all_reduce_sum_(t)
G1 = grads(every, 1.0 / len(every), 1.0)  # one process on the multiset of the global ragged batch (the pad link twice)
scale = np.abs(G1).max()
assert np.abs(G - G1).max() < 2e-5 * scale, np.abs(G - G1).max() / scale
parallel.barrier()
print('rank', rank, 'ragged ok')
'''


def test_data_parallel_ragged_last_batch_gloo(tmp_path):
    """The last batch of an epoch under data parallelism: shards are padded to equal lengths (wrap-around), every rank
    steps on the same, smaller batch size B' and scales by 1 / (B' * world): the all-reduced gradient is the exact mean
    over the global ragged batch (the wrapped link counted where it was dealt)."""
    from helpers import emu_lib
    emu_lib()
    script = tmp_path / 'dpr.py'
    script.write_text(_DP_RAGGED_WORKER % (ROOT, ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29617')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ragged ok' % r in o


def test_host_cpu_budget_and_thread_limit(tmp_path):
    """igmc_amd.hostcpu: the CPU budget honours a cgroup quota (here: whatever this container has), the thread limit is set
    before numpy / torch are imported and never overrides a value the user exported."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from igmc_amd import hostcpu\n"
            "assert 'torch' not in sys.modules and 'numpy' not in sys.modules\n"
            "n = hostcpu.cpu_budget(); assert 1 <= n <= (os.cpu_count() or 1)\n"
            "t = hostcpu.limit_host_threads(); assert 1 <= t <= 4 and os.environ['OMP_NUM_THREADS'] == str(t)\n"
            "os.environ['OMP_NUM_THREADS'] = '7'; assert hostcpu.limit_host_threads() == 7\n"
            "for k in list(os.environ):\n"
            "    if k.endswith('_NUM_THREADS'): del os.environ[k]\n"
            "os.environ['LOCAL_WORLD_SIZE'] = '8'      # eight ranks share the node's budget\n"
            "t8 = hostcpu.limit_host_threads(); assert 1 <= t8 <= max(1, n // 32)\n"
            "print('ok', n, t, t8)\n" % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.endswith('_NUM_THREADS')}
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0 and r.stdout.decode().startswith('ok'), r.stdout.decode()


_DP_PIPELINE_WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, torch
from igmc_amd import _lib, engine, parallel, stepgraph
import parity_checks as PC
from helpers import load_extract_golden
rank, world = parallel.init_from_env('gloo')
MODE = os.environ['DP_MODE']              # flat: gradient kernels -> flat all-reduce -> igmc_step_finish (three calls);
                                          # inside / inside_layers: igmc_train_step_dp (subgraph kernel / per-layer kernels)
be = PC.EmuBackend()                      # kernel logic on the CPU emulation (test infrastructure)
lib = be.lib
case = load_extract_golden()['synth_cap']
lu = case['links'][:, 0].astype(np.int32).copy()
lv = case['links'][:, 1].astype(np.int32).copy()
ly = case['class_values'][case['link_labels']].astype(np.float32)
n_all, B, M, ARR, lr, seed, step0 = len(lu), 2, 1, 0.001, 1e-3, 7, 11
graph = engine.Graph(case['A'], lib=lib)
vp = lambda a: C.c_void_p(a.ctypes.data)


class EmuPipeline(stepgraph.GroupPipeline):
    """The product's host logic (igmc_amd.stepgraph.GroupPipeline: groups, pairs, re-grouping, remainder, ragged batch)
    over the emulation build of the kernels, with the data-parallel step of StepGraph: gradient kernels -> ONE flat
    all-reduce -> igmc_step_finish (Adam + loss + tick)."""

    def __init__(self):
        self._init_pipeline(B, M, use_graph=False)
        self.ctrl = np.zeros(_lib.CTRL['WORDS'], np.int64)
        self.sets = [[], []]
        probe = engine.Batch(graph, B, 1, case['mnph'])
        self.ws = engine.ModelWorkspace(lib, 0, 5, 4, 4, 0, probe.node_capacity, probe.edge_capacity, B)
        lib.call('igmc_model_set_ctrl', self.ws.handle, vp(self.ctrl))
        self.P = PC.flatten_params(self.ws, PC.make_ref_model(4, 5, seed=4))
        self.M1, self.M2, self.G = np.zeros_like(self.P), np.zeros_like(self.P), np.zeros_like(self.P)
        self.out, self.loss, self.total = np.zeros(B, np.float32), np.zeros(2, np.float32), np.zeros(1, np.float64)
        self.collectives = 0
        self.hints = 0
        self.spans = set()

        def host_sum(ptr, n, stream):          # sum over the ranks of the n floats at ptr, in place
            buf = np.ctypeslib.as_array((C.c_float * n).from_address(ptr))
            parallel.all_reduce_sum_(torch.from_numpy(buf))
            self.collectives += 1
            self.spans.add(n)
        self.comm = parallel.HostComm(lib, host_sum, rank, world)
        assert self.comm.info() == (rank, world)

    def _arena(self, q, i):
        while len(self.sets[q]) <= i:
            a = engine.Batch(graph, B, 1, case['mnph'])
            lib.call('igmc_batch_set_ctrl', a.handle, vp(self.ctrl))
            self.sets[q].append(a)
        return self.sets[q][i]

    def _extract(self, arena, sel, nb):
        arena.extract(lu, lv, ly, self.perm, sel, nb, 1.0, seed, 999)

    def _enqueue_step(self, arena, nb):
        if MODE != 'flat':
            # the product's data-parallel step (igmc_train_step_dp): the exchange INSIDE the step, here over gloo through a
            # host-callback communicator (igmc_comm_create_host)
            lib.call('igmc_train_step_dp', self.ws.handle, self.comm.handle, vp(self.P), arena.handle, 0, None, seed, 0, 1.0,
                     ARR, vp(self.out), vp(self.G), vp(self.M1), vp(self.M2), vp(self.loss), vp(self.total), vp(self.ctrl),
                     1, lr, 0.9, 0.999, 1e-8, 0.0, None)
            return
        self.ws.loss_grad(self.P.ctypes.data, arena, self.out.ctypes.data, self.G.ctypes.data, None, seed=seed, step=0,
                          ARR=ARR, grad_scale=1.0 / (nb * world), arr_scale=1.0 / world)
        parallel.all_reduce_sum_(torch.from_numpy(self.G))           # in place on the numpy buffer
        self.collectives += 1
        lib.call('igmc_step_finish', self.ws.handle, arena.handle, vp(self.P), vp(self.G), vp(self.M1), vp(self.M2), ARR,
                 vp(self.loss), vp(self.total), vp(self.ctrl), 1, lr, 0.9, 0.999, 1e-8, 0.0, None)

    def _dev_regroup(self, first_cur, first_next):
        lib.call('igmc_ctrl_regroup', vp(self.ctrl), self.M, first_cur, first_next, None)

    def _hint_unchanged(self):             # (StepGraph's: inside a pair of groups the parameters are the previous step's)
        lib.call('igmc_model_weights_unchanged', self.ws.handle, 1)
        self.hints += 1

    def _fork(self):
        pass

    def _side(self, fn):
        fn()

    def _join(self):
        pass

    def _capture(self):
        return None

    def _count(self, n):
        pass

    def run_epoch(self, perm, epoch):
        pad = 4 * stepgraph.MAX_GROUP * B
        self.perm = np.concatenate([perm, perm[np.arange(pad) %% len(perm)]]).astype(np.int32)
        self.ctrl[:] = stepgraph._ctrl_words(step0, epoch, 1, B, self.M, lr, 0.9, 0.999, 1e-8, 0.0)
        self._reset_epoch(len(perm))
        self._fill_group()
        self.steps(len(perm) // B)
        if len(perm) %% B:
            self.step(len(perm) %% B)


perm_all = torch.randperm(n_all, generator=torch.Generator().manual_seed(3))[:14]
mine = parallel.shard_positions(perm_all, rank, world, pad=True).numpy().astype(np.int32)
assert len(mine) == 7                      # 14 links over 2 ranks, batch 2: a PAIR of groups (2 steps), a remainder step
                                           # launched on its own, and a ragged last batch of 1 link per rank
pipe = EmuPipeline()
pipe.run_epoch(mine, 3)
K = _lib.CTRL
assert pipe.ctrl[K['SYNC_ERR']] == 0 and pipe.ctrl[K['K']] == 4
assert pipe.hints == 1                     # one pair of groups of M = 1: its second step rides on the first one's images
n_lin = 128 * 256 + 128 + 128 + 1
if MODE == 'flat':
    assert pipe.collectives == 4
else:
    # two spans per step: the step's reduced gradient sources (tables + d att partials, or basis-space sums) and lin1 / lin2
    assert pipe.collectives == 8 and len(pipe.spans) == 2 and n_lin in pipe.spans, (pipe.collectives, pipe.spans)
    ts = (5 * 32 + 33) * 32
    src = 4 * ts + 4 * ts // 32 * 4 if MODE == 'inside' else max(pipe.spans - {n_lin})
    assert src in pipe.spans and (MODE == 'inside' or src < 4 * ts), pipe.spans
# replicas stay bit-identical: every rank applied the same all-reduced gradients
both = [torch.zeros(len(pipe.P)) for _ in range(world)]
torch.distributed.all_gather(both, torch.from_numpy(pipe.P.copy()))
assert torch.equal(both[0], both[1])
tot = torch.tensor([float(pipe.total[0])], dtype=torch.float64)
parallel.all_reduce_sum_(tot)
assert np.isfinite(tot.item()) and tot.item() > 0
if rank == 0:
    # the same epoch by direct C-ABI calls with host arguments (no control block, no pipeline): step t = both ranks'
    # batches t one after the other, gradients summed, Adam -- what the pipeline must have walked
    ws = pipe.ws
    lib.call('igmc_model_set_ctrl', ws.handle, None)
    b = engine.Batch(graph, B, 1, case['mnph'])
    P = PC.flatten_params(ws, PC.make_ref_model(4, 5, seed=4))
    M1, M2 = np.zeros_like(P), np.zeros_like(P)
    shards = [parallel.shard_positions(perm_all, r, world, pad=True).numpy().astype(np.int32) for r in range(world)]
    out = np.zeros(B, np.float32)
    for t, (first, nb) in enumerate(((0, 2), (2, 2), (4, 2), (6, 1))):
        Gs = []
        for r in range(world):
            G = np.zeros_like(P)
            b.extract(lu, lv, ly, shards[r], first, nb, 1.0, seed, 3)
            ws.loss_grad(P.ctypes.data, b, out.ctypes.data, G.ctypes.data, None, seed=seed, step=step0 + t, ARR=ARR,
                         grad_scale=1.0 / (nb * world), arr_scale=1.0 / world)
            Gs.append(G)
        G = Gs[0] + Gs[1]
        ws.adam_step(P.ctypes.data, G.ctypes.data, M1.ctypes.data, M2.ctypes.data, t + 1, lr)
    # (lr is a float argument there and a double in the control block: last-ulp differences only)
    np.testing.assert_allclose(pipe.P, P, rtol=2e-5, atol=1e-7)
parallel.barrier()
print('rank', rank, 'pipeline ok')
'''


@pytest.mark.parametrize('mode', ['inside', 'inside_layers', 'flat'])
def test_data_parallel_group_pipeline_gloo(tmp_path, mode):
    """StepGraph's host logic (GroupPipeline) under data parallelism, two ranks over gloo, kernels on the emulator: a pair of
    groups, the remainder and the ragged last batch of an epoch.  `inside`: the product's step, igmc_train_step_dp -- the
    subgraph kernel's tables (`inside_layers`: the per-layer path's basis-space sums) and the lin gradients summed over the
    ranks between their reduction and the gradient / Adam kernel, through a host-callback communicator; `flat`: gradient
    kernels, ONE flat all-reduce, Adam / tick kernel as three calls.  Replicas end bit-identical, no stamp mismatch, the
    collective counts and spans agree, and the result equals the same epoch walked by direct host-argument calls."""
    from helpers import emu_lib
    emu_lib()
    script = tmp_path / 'dpp.py'
    script.write_text(_DP_PIPELINE_WORKER % (ROOT, ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(29617 + ['inside', 'inside_layers', 'flat'].index(mode)), DP_MODE=mode,
                   IGMC_GRAPH_STEP='0' if mode == 'inside_layers' else '1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert 'rank %d pipeline ok' % r in o


def test_group_sizes_of_short_and_odd_runs():
    """Launch structure chosen for a run of a known length: whole pairs of groups where 2 M divides it, ONE single-group launch
    for a run of at most 50 steps, and for an evaluation pass the M that leaves the fewest steps outside whole launches."""
    from igmc_amd.stepgraph import MAX_GROUP, _group_size, _group_size_for
    assert MAX_GROUP == 50
    assert _group_size(200, 50) == 50 and _group_size(400, 50) == 50 and _group_size(64, 50) == 32 and _group_size(128, 50) == 32
    assert _group_size(20, 50) == 20 and _group_size(30, 50) == 30 and _group_size(4, 50) == 4      # single-group launches
    assert _group_size(40, 50) == 40 and _group_size(50, 50) == 50
    assert _group_size(3, 50) == 50 and _group_size(67, 50) == 50                                  # nothing divides: the default
    for n in (16, 17, 24, 53, 100, 273, 2000, 2001):
        cap = min(32, n // 2)
        m = _group_size_for(n)
        assert 8 <= m <= cap
        assert n % (2 * m) == min(n % (2 * k) for k in range(8, cap + 1)), n      # the fewest steps outside whole launches
    assert _group_size_for(8) == 4 and _group_size_for(9) == 4                      # (below sixteen steps: half the run)
    assert _group_size_for(100) == 25 and _group_size_for(24) == 12


def test_pacing_policy_by_the_kernels_of_a_step():
    """Which configurations hold their extraction launches back behind device-side gates, and when a gate opens: the subgraph
    kernel 10 us into the step (training; evaluation passes run their chain free), the group-split dense-layer kernels 40 us in (training only: their
    second launch is on the chip by then), cap-200 arenas and the per-layer kernels run the extraction chain free."""
    from igmc_amd.stepgraph import StepGraph
    assert StepGraph.pacing_policy(1, True) == (True, 10.0) and StepGraph.pacing_policy(1, False)[0] is False
    assert StepGraph.pacing_policy(3, True) == (True, 40.0) and StepGraph.pacing_policy(3, False)[0] is False
    assert StepGraph.pacing_policy(2, True)[0] is False and StepGraph.pacing_policy(0, True)[0] is False


def test_batches_per_extraction_launch_follow_the_density_of_the_graph(monkeypatch):
    """``StepGraph._chunk``: two batches per extraction launch, one where a link's two neighbourhoods hold fewer than 128
    candidates on average (the Monti sets); the environment overrides; evaluation passes keep two."""
    import types
    from igmc_amd.stepgraph import EvalGraph, StepGraph
    monkeypatch.delenv('IGMC_GROUP_EXTRACT_CHUNK', raising=False)

    def chunk(cls, n_users, n_items, nnz, M=25):
        sg = object.__new__(cls)
        sg.M = M
        sg.ds = types.SimpleNamespace(graph=types.SimpleNamespace(n_users=n_users, n_items=n_items, nnz=nnz))
        return sg._chunk()

    assert chunk(StepGraph, 3000, 3000, 26173) == 1            # flixster: 17 candidates a link
    assert chunk(StepGraph, 3000, 3000, 123202) == 1           # douban: 82
    assert chunk(StepGraph, 943, 1682, 100000) == 2            # ml_100k: 165
    assert chunk(StepGraph, 6040, 3706, 1000209) == 2          # ml_1m: 435
    assert chunk(EvalGraph, 3000, 3000, 26173) == 2
    assert chunk(StepGraph, 6040, 3706, 1000209, M=1) == 1     # (never more than a group)
    monkeypatch.setenv('IGMC_GROUP_EXTRACT_CHUNK', '4')
    assert chunk(StepGraph, 3000, 3000, 26173) == 4

