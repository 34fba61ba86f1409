import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the library reads ORBX_MATCH_FORM once per process unless this is set before its first call: tests switch the form
os.environ.setdefault("ORBX_SWITCHES_PER_CALL", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle(1000, 1.2, 8, 20, 7)


@pytest.fixture(scope="session")
def euroc_left():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_l.npy"))


@pytest.fixture(scope="session")
def euroc_right():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "fixtures", "EuRoC_r.npy"))
