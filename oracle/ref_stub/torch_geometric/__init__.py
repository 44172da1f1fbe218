"""Three-class stand-in for ``torch_geometric`` (TEST INFRASTRUCTURE ONLY).

The reference's ``util_functions.py`` (``/root/reference/util_functions.py:13``)
needs only the *names* ``Data, Dataset, InMemoryDataset`` at import time.  This
stub lets ``tests/golden/make_golden.py`` import the unmodified reference
extractor inside the build container to generate golden vectors.  It is never
imported by the product package.
"""
