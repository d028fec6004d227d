"""Parametrised fixtures the reference's tests expect (test harness only; see tests/refcompat/README.md)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from sourmash_tst_utils import RunnerContext, TempDirectory   # noqa: E402


@pytest.fixture
def runtmp():
    with TempDirectory() as location:
        yield RunnerContext(location)


@pytest.fixture
def run():
    yield RunnerContext(os.getcwd())


def _bool_fixture(name):
    @pytest.fixture(params=[True, False], name=name)
    def fx(request):
        return request.param
    return fx


track_abundance = _bool_fixture("track_abundance")
dayhoff = _bool_fixture("dayhoff")
hp = _bool_fixture("hp")
keep_identifiers = _bool_fixture("keep_identifiers")
keep_versions = _bool_fixture("keep_versions")
use_manifest = _bool_fixture("use_manifest")


@pytest.fixture(params=[2, 5, 10])
def n_children(request):
    return request.param


@pytest.fixture(params=["--linear", "--no-linear"])
def linear_gather(request):
    return request.param


@pytest.fixture(params=["--prefetch", "--no-prefetch"])
def prefetch_gather(request):
    return request.param


def pytest_addoption(parser):
    parser.addoption("--run-hypothesis", action="store_true", help="run hypothesis tests")


def pytest_configure(config):
    config.addinivalue_line("markers", "hypothesis: property tests")
