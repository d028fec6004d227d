"""Helper names the reference's tests import (test harness only; see tests/refcompat/README.md)."""
import os
import shutil
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "test-data")               # staged next to this file by tools/run_reference_tests.sh

SIG_FILES = [os.path.join("demo", f) for f in
             ("SRR2060939_1.sig", "SRR2060939_2.sig", "SRR2241509_1.sig", "SRR2255622_1.sig", "SRR453566_1.sig",
              "SRR453569_1.sig", "SRR453570_1.sig")]


def get_test_data(filename):
    return os.path.join(DATA, filename)


class SourmashCommandFailed(Exception):
    def __init__(self, msg=""):
        Exception.__init__(self, msg)
        self.message = msg


class TempDirectory:
    def __init__(self):
        self.tempdir = tempfile.mkdtemp(prefix="sourmashtest_")

    def __enter__(self):
        return self.tempdir

    def __exit__(self, exc_type, exc_value, traceback):
        shutil.rmtree(self.tempdir, ignore_errors=True)
        return False


class RunnerContext:
    "Holds a scratch location; anything that would drive the command line is out of scope and skips."

    def __init__(self, location):
        self.location = location
        self.last_command = self.last_result = None

    def output(self, path):
        return os.path.join(self.location, path)

    def run_sourmash(self, *args, **kwargs):
        pytest.skip("command line is out of scope (SURVEY.md section 8)")

    sourmash = run = run_sourmash

    def __str__(self):
        return f"RunnerContext({self.location})"


def in_tempdir(fn):
    def wrapper(*args, **kwargs):
        with TempDirectory() as location:
            return fn(RunnerContext(location), *args, **kwargs)
    wrapper.__name__ = fn.__name__
    return wrapper


def in_thisdir(fn):
    def wrapper(*args, **kwargs):
        return fn(RunnerContext(os.getcwd()), *args, **kwargs)
    wrapper.__name__ = fn.__name__
    return wrapper


def runscript(*args, **kwargs):
    pytest.skip("command line is out of scope (SURVEY.md section 8)")
