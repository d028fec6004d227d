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


def _choice_fixture(name, choices):
    @pytest.fixture(params=choices, name=name)
    def fx(request):
        return request.param
    return fx


lca_db_format = _choice_fixture("lca_db_format", ["json", "sql"])
manifest_db_format = _choice_fixture("manifest_db_format", ["csv", "sql"])
sig_save_extension = _choice_fixture("sig_save_extension", ["sig", "sig.gz", "zip", ".d/", ".sqldb"])
sig_save_extension_abund = _choice_fixture("sig_save_extension_abund", ["sig", "sig.gz", "zip", ".d/"])
abspath_or_relpath = _choice_fixture("abspath_or_relpath", ["--abspath", "--relpath"])
abspath_relpath_v4 = _choice_fixture("abspath_relpath_v4", ["--no-abspath", "--abspath", "--relpath"])


def pytest_addoption(parser):
    parser.addoption("--run-hypothesis", action="store_true", help="run hypothesis tests")


def pytest_configure(config):
    config.addinivalue_line("markers", "hypothesis: property tests")
