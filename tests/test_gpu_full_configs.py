"""BASELINE.json configurations C4 (compare, N = 10,000) and C5 (gather, 10^6-hash query vs 100,000 sketches) at
FULL size on the GPU.  The oracle cannot finish these in test time, so the results are tied down through
size-independent properties (the drivers in tools/ compute them; each is listed in their `checks` dict):
compare -- two independent kernels (merge walk, bit rows) agree bit for bit, symmetry, diagonal = sizes,
planted duplicate / disjoint / superset rows, Jaccard = one IEEE divide; gather -- winners distinct, overlaps
non-increasing and >= threshold, sum |I| = covered query hashes, final counters = an independent streaming
recount, stop rule, round 0 = arg-max with the lowest-index tie-break.  Small versions of the same generators
are compared with the oracle exactly in test_gpu_compare.py / test_gpu_gather.py / test_gpu_parallel.py."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    p = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert lines, p.stdout[-2000:] + p.stderr[-2000:]
    return p.returncode, json.loads(lines[-1])


def test_compare_c4_full_size():
    rc, out = _run([os.path.join("tools", "bench_compare.py"), "c4"])
    assert out["config"]["pairs"] == 49_995_000
    assert all(out["checks"].values()), out["checks"]
    assert rc == 0


def test_gather_c5_full_size():
    rc, out = _run([os.path.join("tools", "bench_gather.py"), "--stepwise"])
    assert out["config"]["datasets"] == 100_000 and out["config"]["query_hashes"] == 1_000_000
    assert all(out["checks"].values()), out["checks"]
    assert out["stepwise_identical"] and out["rounds"] > 1000
    assert rc == 0
