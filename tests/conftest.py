import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# a fatal signal inside the library leaves its native backtrace on stderr before pytest's fault handler prints the Python one
# (round 5: one SIGSEGV inside a host index build, once, in a GPU run that had left nothing but the Python stack)
os.environ.setdefault('PSIGPU_SEGV_TRACE', '1')

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REF_DATA = os.path.join(GOLDEN, 'ref_data')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def ref_data():
    return REF_DATA


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
