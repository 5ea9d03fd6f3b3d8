import os
import sys

import pytest

# MIOpen's immediate mode may otherwise fall back to its naive reference solvers for float32
# convolutions at 24 x 640x480 (minutes per call); bench.py sets the same switches
for _k in ('FWD', 'BWD', 'WRW'):
    os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_' + _k, '0')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
