"""Stand-in for ``torch_geometric`` (TEST INFRASTRUCTURE ONLY; never imported by the product package).

Lets the UNMODIFIED reference modules be imported inside the build container so that golden vectors can be
generated from the reference's own code:

* ``data``   -- ``Data, Dataset, InMemoryDataset`` (names needed by ``/root/reference/util_functions.py:13``) and
  ``DataLoader, DenseDataLoader`` (``/root/reference/train_eval.py:12``): attribute storage + the PyG collate.
* ``nn``     -- ``RGCNConv, GCNConv, global_sort_pool, global_add_pool`` (``/root/reference/models.py:6``).
* ``utils``  -- ``dropout_adj`` (``/root/reference/models.py:7``) with an injectable / recorded mask.

``torch_geometric==1.4.2`` itself (``/root/reference/README.md:26``) is absent: the three operators
(``RGCNConv``, ``dropout_adj``, ``global_sort_pool``) and the collate are bound to ``oracle/pyg_ref``'s restatements
of the published algorithm -- those stay unpinned.  Everything the reference's OWN files compute around them
(``models.py:123-217``, ``train_eval.py:149-245``) is what ``tests/golden/make_model_golden.py`` pins.
"""
