"""Sanitizer runs: the host side of the C-ABI (landing pads, sketch / signature containers,
JSON, md5, the zip / manifest loader; `make -C sourmash_amd/csrc asan`) and the CPU oracle (`make -C oracle asan`) built
with AddressSanitizer + UndefinedBehaviorSanitizer; their CPU test suites then run in a subprocess with the sanitizer
runtime preloaded.  Any report aborts that process (-fno-sanitize-recover, ASan's default abort on error)."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

ASAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _run(preload, extra_env, tests):
    env = dict(os.environ, LD_PRELOAD=preload, **ASAN_ENV, **extra_env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + tests, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "passed" in p.stdout and "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
    return p.stdout


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_c_abi_host_side_under_asan_ubsan():
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("clang's shared ASan runtime is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "sourmash_amd", "csrc"), "-j8", "-s", "asan"])
    lib = os.path.join(ROOT, "sourmash_amd", "libsourmash_amd_asan.so")
    out = _run(rt[0], {"SMG_LIBRARY": lib}, ["tests/test_capi_cpu.py", "tests/test_collection_cpu.py", "tests/test_fastcall_cpu.py", "tests/test_pargz_cpu.py"])
    assert "failed" not in out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_under_asan_ubsan():
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(rt):
        pytest.skip("gcc's ASan runtime is not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    out = _run(rt, {"ORACLE_LIBRARY": lib}, ["tests/test_oracle.py"])
    assert "failed" not in out
