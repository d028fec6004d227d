import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

collect_ignore = ["refcompat", "check_c2_full.py"]          # harness for the reference's own tests / a standalone script


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(*parts):
    return os.path.join(GOLDEN, *parts)


@pytest.fixture(scope="session")
def golden_path():
    return golden
