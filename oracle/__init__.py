"""CPU oracle for the IGMC hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from this package, and only as the checker /
the timed CPU baseline.  The product package ``igmc_amd`` never imports it and
fails loudly when its HIP library is missing.

Contents
--------
``extract_ref``   numpy/scipy restatement of the reference's enclosing-subgraph
                  extraction (``/root/reference/util_functions.py:208-304``).
                  PINNED against golden vectors produced by the *unmodified*
                  reference extractor (``tests/golden/make_golden.py``).
``pyg_ref``       pure-torch (CPU, fp32/fp64) restatement of the PyTorch-Geometric
                  1.4.2 operators the reference model is built from (``RGCNConv``,
                  ``dropout_adj``, ``Batch`` collate) plus ``models.IGMC.forward``
                  (``/root/reference/models.py:170-217``) and the train step
                  (``/root/reference/train_eval.py:149-179``).
                  PARITY UNPINNED for the PyG half: torch_geometric 1.4.2 is an
                  un-vendored dependency, absent from ``/root/reference`` and not
                  installable here (no network); the reference holds no tests or
                  golden vectors for it.  The restatement follows the published
                  PyG-1.4.2 algorithm and is anchored on
                  the reference's own call sites and on the in-tree ARR code
                  (``train_eval.py:167-174``) that pins parameter names/shapes and
                  ``W = att @ basis.view(num_bases, -1)``.
"""
