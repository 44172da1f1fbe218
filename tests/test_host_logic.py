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
