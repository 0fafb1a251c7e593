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
# (Round 5 pinned every test process to the bit-exact routing kernels with XH_ROUTE_REASSOC=0.  Since round 6 the test processes
# run what the library ships -- the reassociated form, on prepared plans where the caller holds velocities and lengths -- and
# hold it to that form's bar; tests that assert BITS ask for the bit-exact kernels by flag: XH_ROUTE_EXACT, which wins over
# the default and the environment.)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
