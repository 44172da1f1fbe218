"""Kernel LOGIC of the HIP extraction (igmc_amd/csrc/extract.hip) on the CPU: the same sources are
compiled against tools/hipemu/hipemu.h and driven through the same C ABI.  The real gfx950 binary is
checked by tests/test_gpu_*.py with identical assertions."""
import numpy as np
import pytest

import parity_checks as PC
from helpers import load_extract_golden

CASES = load_extract_golden()


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


@pytest.mark.parametrize('name', ['hand', 'hand_h2', 'flixster', 'yahoo_music', 'flixster_h2', 'synth_nocap'])
def test_free_run_matches_reference(be, name):
    case = dict(CASES[name])
    if name in ('flixster', 'yahoo_music'):
        case['recs'] = case['recs'][:10]
        case['links'] = case['links'][:10]
        case['link_labels'] = case['link_labels'][:10]
    g, b, d = PC.extract_case(be, case, replay=False)
    PC.check_against_golden(d, case)


@pytest.mark.parametrize('name', ['douban_cap20', 'synth_cap', 'synth_h2_ratio'])
def test_replay_matches_reference(be, name):
    case = dict(CASES[name])
    case['recs'] = case['recs'][:8]
    case['links'] = case['links'][:8]
    case['link_labels'] = case['link_labels'][:8]
    g, b, d = PC.extract_case(be, case, replay=True)
    PC.check_against_golden(d, case)


@pytest.mark.parametrize('name', ['synth_cap', 'synth_h2_ratio'])
def test_sampler_free_run(be, name):
    case = dict(CASES[name])
    g, b, d = PC.extract_case(be, case, replay=False, seed=5, epoch=1)
    PC.check_sampled(d, case)
    # deterministic for a fixed (seed, epoch); a different epoch re-samples
    _, _, d2 = PC.extract_case(be, case, replay=False, seed=5, epoch=1)
    assert np.array_equal(d['node_gid'], d2['node_gid']) and np.array_equal(d['col'], d2['col'])
    _, _, d3 = PC.extract_case(be, case, replay=False, seed=5, epoch=2)
    assert not np.array_equal(d['node_gid'], d3['node_gid'])


def test_sampler_four_pass_path_keeps_the_same_subset(be, monkeypatch):
    """sample_fringe parks the keys of the deciding byte in a 256-entry list and falls back to radix_select's four passes
    when they would not fit -- which a hash never produces.  With a bound of 0 (emulation builds read it from the environment)
    the fallback is taken on every draw and must extract the very same batches."""
    for name in ('synth_cap', 'synth_h2_ratio'):
        case = dict(CASES[name])
        monkeypatch.delenv('IGMC_SAMPLE_PARK', raising=False)
        _, _, d = PC.extract_case(be, case, replay=False, seed=7, epoch=3)
        monkeypatch.setenv('IGMC_SAMPLE_PARK', '0')
        _, _, d2 = PC.extract_case(be, case, replay=False, seed=7, epoch=3)
        PC.check_sampled(d2, case)
        for k in ('node_gid', 'col', 'row_ptr', 'etype'):
            if k in d:
                assert np.array_equal(d[k], d2[k]), (name, k)


def test_arena_overflow_is_reported(be):
    from igmc_amd import engine
    case = CASES['synth_nocap']
    g = engine.Graph(case['A'], lib=be.lib)
    b = engine.Batch(g, max_graphs=2, hop=1, max_nodes_per_hop=None)
    with pytest.raises(RuntimeError):
        lu = np.zeros(3, np.int32)
        b.extract(lu, lu, np.zeros(3, np.float32), None, 0, 3)


def test_cached_node_sets_rebuild_the_same_batches():
    """``igmc_extract_batch_cached`` (static dataset, reference MyDataset): node sets + hop distances kept per link in
    packed arrays; any batch (permuted positions) rebuilt from them equals the free-running extraction of those links."""
    import numpy as np
    import parity_checks as PC
    from helpers import load_extract_golden
    from igmc_amd import engine
    be = PC.EmuBackend()
    for name in ('synth_cap', 'flixster_h2', 'synth_nocap'):
        case = load_extract_golden()[name]
        n = len(case['links'])
        g = engine.Graph(case['A'], lib=be.lib)
        lu = case['links'][:, 0].astype(np.int32).copy()
        lv = case['links'][:, 1].astype(np.int32).copy()
        ly = case['class_values'][case['link_labels']].astype(np.float32)
        b = engine.Batch(g, n, case['h'], case['mnph'])
        b.extract(lu.ctypes.data, lv.ctypes.data, ly.ctypes.data, None, 0, n, sample_ratio=case['sample_ratio'], seed=4, epoch=0)
        d = b.download()
        uoff, voff = np.zeros(n + 1, np.int64), np.zeros(n + 1, np.int64)
        un, vn, ud, vd = [], [], [], []
        for k in range(n):
            lo, hi, nu = d['node_off'][k], d['node_off'][k + 1], d['n_users'][k]
            un.append(d['node_gid'][lo:lo + nu]); vn.append(d['node_gid'][lo + nu:hi])
            ud.append(d['node_label'][lo:lo + nu] // 2); vd.append(d['node_label'][lo + nu:hi] // 2)
            uoff[k + 1], voff[k + 1] = uoff[k] + nu, voff[k] + (hi - lo - nu)
        arrs = dict(uoff=uoff, voff=voff, unodes=np.concatenate(un).astype(np.int32), vnodes=np.concatenate(vn).astype(np.int32),
                    udist=np.concatenate(ud).astype(np.uint8), vdist=np.concatenate(vd).astype(np.uint8))
        cache = {k: v.ctypes.data for k, v in arrs.items()}
        perm = np.random.default_rng(1).permutation(n).astype(np.int32)
        B = min(4, n - 1)
        b2 = engine.Batch(g, B, case['h'], case['mnph'])
        b2.extract_cached(cache, ly.ctypes.data, perm.ctypes.data, 1, B)
        d2 = b2.download()
        b3 = engine.Batch(g, B, case['h'], case['mnph'])
        b3.extract(lu.ctypes.data, lv.ctypes.data, ly.ctypes.data, perm.ctypes.data, 1, B, sample_ratio=case['sample_ratio'], seed=4, epoch=0)
        d3 = b3.download()
        for key in ('node_off', 'n_users', 'node_label', 'node_gid', 'row_ptr', 'col', 'erel', 'y'):
            assert np.array_equal(d2[key], d3[key]), (name, key)
