import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sf1():
    """Official-dbgen-equivalent SF1 tables (tests/golden pins them), generated once."""
    import tpch_data

    return tpch_data.load(1, 1, text=True)


@pytest.fixture(scope="session")
def sf001():
    import tpch_data

    return tpch_data.load(1, 100)
