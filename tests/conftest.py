import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The per-box cache of learnt cells (xh_route_plan_prepare) makes a plan's FIRST call depend on what earlier plans of the same
# grid did on this machine -- right for a product run, wrong for tests that assert which form a call ran in.  Off unless a test
# (in a child process) switches it on.
os.environ.setdefault('XH_ROUTE_LEARN_CACHE', '0')
# Since round 5 the library routes tree networks in the reassociated form by default (equal to the reference to rounding, not
# bit for bit).  The suites written against the bit-exact kernels -- which stay the checker -- keep asserting bits, so the
# default of the test processes is the bit-exact form; tests/test_gpu_reassoc.py, the full-size reassociated test and the
# default-path tests ask for the other form by flag (a call's flag wins over the environment) or in a child process.
os.environ.setdefault('XH_ROUTE_REASSOC', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
