"""The bench line is a record a driver parses: ONE line of strict JSON, at most 8 KB, with the contract keys
(a 27 KB line came back as `parsed: null`).  Canned inputs, no GPU."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def canned(bench, n_ranks=1):
    "a line as main() assembles it, from the committed round-5 record (its `extra` feeds the summary, as in the run)"
    doc = json.load(open(os.path.join(ROOT, "profiles", "r05_bench.json")))
    extra = doc.pop("extra")
    doc["summary"] = bench.extras_summary(extra)
    doc["extra_file"] = bench.EXTRA_FILE
    doc["config"]["comm"] = {"backend": "nccl", "world_size_observed": n_ranks, "library": "RCCL 2.26.6",
                             "devices": ["AMD Instinct MI355X #%d" % i for i in range(n_ranks)],
                             "ranks_share_one_gpu": False, "distinct_devices": n_ranks}
    return doc, extra


@pytest.mark.parametrize("n_ranks", [1, 8])
def test_line_is_small_strict_and_complete(bench, n_ranks):
    doc, _ = canned(bench, n_ranks)
    line = bench.finalize_line(doc)
    assert "\n" not in line
    assert len(line.encode()) < 8192
    back = json.loads(line, parse_constant=lambda c: pytest.fail("non-strict constant %s" % c))
    for k in bench.REQUIRED_KEYS + ("summary",):
        assert k in back, k
    assert "extra" not in back                               # the 22 KB of secondary metrics live in the side file
    assert "shed" not in back                                # nothing had to go
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["config"]["workload"]


def test_nan_and_infinity_become_null(bench):
    doc, _ = canned(bench)
    doc["roofline"]["frac"] = float("nan")
    doc["summary"]["c3_compare_1000_auto_ms"] = float("inf")
    line = bench.finalize_line(doc)
    assert "NaN" not in line and "Infinity" not in line
    back = json.loads(line)
    assert back["roofline"]["frac"] is None and back["summary"]["c3_compare_1000_auto_ms"] is None


def test_an_oversized_line_sheds_prose_not_contract_fields(bench):
    doc, _ = canned(bench)
    doc["roofline"]["traffic_from"] = "x" * 6000
    doc["cpu_baseline"]["note"] = "y" * 6000
    line = bench.finalize_line(doc)
    assert len(line.encode()) <= bench.LINE_LIMIT
    back = json.loads(line)
    assert "roofline.traffic_from" in back["shed"]
    for k in bench.REQUIRED_KEYS:
        assert k in back
    assert back["roofline"]["frac"] == doc["roofline"]["frac"]


def test_a_line_that_cannot_fit_or_lacks_a_key_is_an_error(bench):
    doc, _ = canned(bench)
    doc["metric"] = "m" * 9000
    with pytest.raises(ValueError):
        bench.finalize_line(doc)
    doc, _ = canned(bench)
    del doc["roofline"]
    with pytest.raises(ValueError):
        bench.finalize_line(doc)


def test_extras_go_to_the_side_file_and_stderr(bench, tmp_path, capfd, monkeypatch):
    _, extra = canned(bench)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.write_extras(extra, {"backend": None})
    out, err = capfd.readouterr()
    assert out == ""                                         # stdout carries the one contract line only
    assert err.startswith("BENCH_EXTRA {")
    side = json.load(open(tmp_path / bench.EXTRA_FILE))
    assert side["extra"].keys() == extra.keys()


def test_default_workload_is_the_literal_c2(bench, monkeypatch):
    monkeypatch.setattr("sys.argv", ["bench.py"])
    args = bench.parse()
    assert args.records == 1000 and args.record_len == 10_000_000 and not args.bases
    assert args.records * (args.record_len + 1) == 10_000_001_000
