import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("SS_RANDOM_INIT", "1")          # no weights exist offline: tests run the seeded random-init networks


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the round-end driver)")


@pytest.fixture(scope="session")
def cfg():
    from strongsort_yolo_amd.config import StrongSortConfig
    return StrongSortConfig()
