"""CPU extraction: the UNMODIFIED reference extractor against the oracle's restatement of it, same links, same machine.

    python tools/exp_ref_vs_port_cpu.py [n_links]          (build container only: needs /root/reference)

bench.py's ``cpu_baseline`` runs on the GPU box, where /root/reference does not exist and reference sources may not be copied
into the repository -- so it times ``oracle/extract_ref.py`` (kind "port").  This script pins what that substitution is worth:
it imports ``/root/reference/util_functions.py`` through the 3-class ``torch_geometric`` stub (``oracle/ref_stub``) and times
its ``subgraph_extraction_labeling`` + ``construct_pyg_graph`` (reference :208-297) against ``extract_ref.extract`` on the
same links of the ml_1m-shaped graph bench.py uses (hop 1, cap 100), one process, one core.
"""
import os
import random
import sys
import time
import warnings

import numpy as np

warnings.simplefilter('ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_stub'))
sys.path.insert(0, '/root/reference')
sys.path.insert(0, ROOT)
import util_functions as REF  # noqa: E402  (the unmodified reference module)
from igmc_amd import preprocessing  # noqa: E402
from oracle import extract_ref  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    (_, _, A, tr_l, tr_u, tr_v, _, _, _, _, _, _, cv) = preprocessing.create_trainvaltest_split('ml_1m', 1234, True, verbose=False)
    Acsc = A.tocsc()
    Arow, Acol = REF.SparseRowIndexer(A), REF.SparseColIndexer(Acsc)
    pick = np.random.default_rng(0).permutation(len(tr_u))[:n]
    for name, fn in (('reference util_functions.subgraph_extraction_labeling + construct_pyg_graph',
                      lambda k: REF.construct_pyg_graph(*REF.subgraph_extraction_labeling(
                          (tr_u[k], tr_v[k]), Arow, Acol, 1, 1.0, 100, None, None, cv, tr_l[k]))),
                     ('oracle/extract_ref.extract (the restatement bench.py times on the GPU box)',
                      lambda k: extract_ref.extract((tr_u[k], tr_v[k]), A, Acsc, 1, 1.0, 100, cv, tr_l[k]))):
        random.seed(1)
        fn(pick[0])
        t0 = time.perf_counter()
        for k in pick:
            fn(k)
        dt = time.perf_counter() - t0
        print('%-90s %7.1f subgraphs/s  (%d links, %.2f s, 1 core)' % (name, n / dt, n, dt))


if __name__ == '__main__':
    main()
