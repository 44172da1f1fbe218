import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    # same-box A/B tooling (tools/build_patch_variants.py): the GPU suite against an experimental build of the library
    if os.environ.get('IGMC_LIB_PATH'):
        from igmc_amd import _lib
        _lib.LIB_PATH = os.environ['IGMC_LIB_PATH']
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: longer CPU test')
