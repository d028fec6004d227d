"""bench.py reads counters (HBM traffic, SQ counters) from the committed rocprofv3 summaries through profiles/pmcfile.py and must
refuse them when the kernel's sources changed after the profile was taken: the parser and the staleness rule, on the host."""
import os
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "profiles"))
import pmcfile  # noqa: E402


def test_committed_summaries_parse():
    p = pmcfile.PmcFile("profiles/r03_pmc.txt")
    assert p.exists and p.sources and "sketch.hip" in p.sources
    k = "sketch_dna_kernel<31, 16, false>"
    fetch, write = p.get(k, "FETCH_SIZE"), p.get(k, "WRITE_SIZE")
    # 2 x FETCH_SIZE + WRITE_SIZE (KiB) = the 10 GB batch read once + 80 MB of kept hashes written
    assert 9.9e9 < (2 * fetch + write) * 1024 < 10.6e9
    assert p.get(k, "SQ_INSTS_VALU") > 1e10 and p.get("no_such_kernel", "FETCH_SIZE") is None
    g = pmcfile.PmcFile("profiles/r03_gather_pmc.txt")
    assert g.get("build_range_kernel<0>", "FETCH_SIZE") > 1e6 and g.get("overlap_wide_kernel", "FETCH_SIZE") > 3e6
    assert g.sum_over(["build_bounds_kernel", "build_partition_kernel"], "WRITE_SIZE") > 0


def test_staleness_rule(tmp_path):
    now = pmcfile.source_hashes()
    assert "gather.hip" in now and len(now["gather.hip"]) == 12
    good = tmp_path / "good.txt"
    good.write_text(pmcfile.header_line() + "\n  smg::k                       FETCH_SIZE        2        10.0     20.0\n")
    p = pmcfile.PmcFile(str(good))
    assert p.stale(["gather.hip", "qindex.hpp"]) is None and p.get("smg::k", "FETCH_SIZE") == 10.0
    bad = tmp_path / "bad.txt"
    bad.write_text(pmcfile.header_line().replace(now["gather.hip"], "0" * 12) + "\n")
    why = pmcfile.PmcFile(str(bad)).stale(["gather.hip"])
    assert why and "gather.hip changed" in why
    assert "no source hashes" in pmcfile.PmcFile("profiles/r02_pmc.txt").stale(["sketch.hip"])
    assert "absent" in pmcfile.PmcFile("profiles/nope.txt").stale(["sketch.hip"])


def test_valu_mix_file_is_current_and_calibration_is_usable():
    """bench.py prices SQ_INSTS_VALU with the instruction mix of profiles/valu_mix_sketch.json (tools/valu_mix.py) and turns
    FETCH_SIZE / WRITE_SIZE into bytes with the ratios of profiles/r04_fetch_calib.txt.  The mix file must belong to the
    present sources of the sketch kernel (bench.py refuses a stale one: regenerate with `python tools/valu_mix.py`), its
    prediction of the instruction count must stay near the counter's, and the calibration must give one ratio per direction."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from pmcfile import Calibration, ValuMix
    mix = ValuMix()
    assert mix.stale() is None, mix.stale()
    assert 3.0 < mix.doc["mix_cycles_per_valu_inst"] < 4.4
    assert abs(mix.doc["expected_valu_insts_per_kmer"] - 116.5) / 116.5 < 0.08        # profiles/r03_pmc.txt: 116.5 per k-mer
    assert not [op for op in mix.doc["unmeasured_opcodes"] if not op.startswith(("v_min", "v_mbcnt", "v_max"))]
    cal = Calibration()
    assert cal.ok and cal.read_ratio == 0.5 and cal.write_ratio == 1.0
    assert cal.bytes_read(1024) == 2 * 1024 * 1024 and cal.bytes_written(1024) == 1024 * 1024
    assert cal.ratios[("read_cursor_b64_kernel", "FETCH_SIZE")] > 0.6            # the overlap pass's access shape over-fetches
