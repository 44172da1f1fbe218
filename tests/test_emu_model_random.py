"""Randomised parity of the HIP model kernels (CPU emulation of igmc_amd/csrc/{model,graphstep}.hip) against the
PyG-1.4.2 restatement (oracle/pyg_ref.py) on random rating graphs: 2-6 relations, 1-2 hops, with and without per-hop
caps (capped cases with <= 5 relations run the one-workgroup-per-subgraph kernel, the others the per-layer kernels),
with and without edge dropout, subgraphs with no edges at all included."""
import pytest

import parity_checks as PC
from helpers import random_case


@pytest.fixture(scope='module')
def be():
    return PC.EmuBackend()


@pytest.mark.parametrize('h,mnph', [(1, None), (1, 6), (2, 4), (1, 12)])
def test_random_graphs_forward_backward(be, monkeypatch, h, mnph):
    monkeypatch.setenv('IGMC_GRAPH_STEP', '1')
    for seed in range(6):
        case = random_case(31000 + 17 * h + seed, h, mnph=mnph, n_links=5)
        R = len(case['class_values'])
        res = PC.run_model_parity(be, case, R=R, use_dropout=bool(seed % 2), multiply_by=1.0 + (seed % 3))
        assert res['worst_grad_err'] < 1e-4
