import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs through the HIP kernels via the C-ABI)")


@pytest.fixture(scope="session")
def params():
    from hunter_bipedal_control_amd import ingest
    return ingest.load_packaged()


@pytest.fixture(scope="session")
def oracle(params):
    from oracle.pyoracle import Oracle
    return Oracle(params)
