#!/usr/bin/env python3
"""Counter summaries under profiles/ as data: bench.py reads HBM traffic and SQ counters of a kernel from the committed
rocprofv3 --pmc summary instead of carrying constants, and refuses them when the kernel's sources have changed since the
profile was taken.

A summary (written by profiles/summarize.py through tools/prof_r03.sh) starts with a header line

    # sources: sketch.hip=1a2b3c4d5e6f kmer_core.hpp=... ...

holding the first 12 hex digits of the SHA-1 of every file under sourmash_amd/csrc at profiling time, followed by the
per-pass tables whose counter rows read `  <kernel>  <counter>  <dispatches>  <avg/dispatch>  <sum>`.

    python profiles/pmcfile.py header            -> prints the header line for the present tree
"""
import hashlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sourmash_amd", "csrc")


def source_hashes():
    out = {}
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".cpp", ".c")):
            with open(os.path.join(CSRC, name), "rb") as f:
                out[name] = hashlib.sha1(f.read()).hexdigest()[:12]
    return out


def header_line():
    return "# sources: " + " ".join(f"{k}={v}" for k, v in source_hashes().items())


class PmcFile:
    def __init__(self, path):
        self.path = path
        self.sources = {}
        self.rows = []          # (kernel, counter, dispatches, avg, sum)
        self.durations = {}     # kernel -> [avg_us of each pass]
        full = path if os.path.isabs(path) else os.path.join(ROOT, path)
        self.exists = os.path.exists(full)
        if not self.exists:
            return
        row = re.compile(r"^\s{2}(\S.*?)\s{2,}([A-Z][A-Za-z0-9_]+)\s+(\d+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s*$")
        stat = re.compile(r"^(\S.*?)\s{2,}(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s")
        with open(full) as f:
            for line in f:
                if line.startswith("# sources:"):
                    self.sources = dict(kv.split("=", 1) for kv in line.split()[2:])
                    continue
                m = row.match(line.rstrip("\n"))
                if m:
                    self.rows.append((m.group(1).strip(), m.group(2), int(m.group(3)), float(m.group(4)), float(m.group(5))))
                    continue
                m = stat.match(line.rstrip("\n"))
                if m and not line.startswith("kernel "):
                    self.durations.setdefault(m.group(1).strip(), []).append(float(m.group(4)))

    def stale(self, files):
        """None if the profile was taken with the present versions of `files`; else the reason it cannot be quoted"""
        if not self.exists:
            return f"{self.path} is absent"
        if not self.sources:
            return f"{self.path} records no source hashes (taken before round 3)"
        now = source_hashes()
        changed = [f for f in files if self.sources.get(f) != now.get(f)]
        return f"{', '.join(changed)} changed since {self.path} was taken" if changed else None

    def get(self, kernel, counter, which="avg"):
        "counter value per dispatch (or summed) of the kernel whose name contains `kernel`; None if absent"
        hits = [r for r in self.rows if kernel in r[0] and r[1] == counter]
        if not hits:
            return None
        if which == "sum":
            return sum(r[4] for r in hits)
        n = sum(r[2] for r in hits)
        return sum(r[3] * r[2] for r in hits) / n if n else None

    def sum_over(self, kernels, counter):
        "sum over several kernels of the per-run total of a counter (kernels that run once per build each)"
        vals = [self.get(k, counter, "sum") for k in kernels]
        return None if any(v is None for v in vals) else sum(vals)


class Calibration:
    """FETCH_SIZE / WRITE_SIZE -> bytes, with the ratios measured on kernels that move a known number of bytes
    (profiles/r04_fetch_calib.txt, written by tools/prof_calib.sh from tools/ubench/fetch_calib.hip): on gfx950 under this
    rocprofv3 FETCH_SIZE reports 0.500 of the bytes of EVERY coalesced read that was measured -- 4, 8 and 16 B per lane,
    `global_load ... lds` alike -- and WRITE_SIZE 1.000 of coalesced writes; a lone 8-byte read counts 64 B, a lone 4-byte
    write 32 B.  `bytes_read(kib)` / `bytes_written(kib)` apply the coalesced ratios; `describe()` says which were used."""
    FILE = "profiles/r04_fetch_calib.txt"
    READ_ROWS = ("read_b32_kernel", "read_b64_kernel", "read_b128_kernel", "read_lds_b32_kernel")
    WRITE_ROWS = ("write_b32_kernel", "write_b64_kernel", "write_b128_kernel")

    def __init__(self, path=None):
        self.path = path or self.FILE
        self.ratios = {}
        full = self.path if os.path.isabs(self.path) else os.path.join(ROOT, self.path)
        if os.path.exists(full):
            with open(full) as f:
                for line in f:
                    if line.startswith("calib "):
                        parts = line.split("|")[0].split()
                        self.ratios[(parts[1], parts[2])] = float(parts[5])
        r = [self.ratios.get((k, "FETCH_SIZE")) for k in self.READ_ROWS]
        w = [self.ratios.get((k, "WRITE_SIZE")) for k in self.WRITE_ROWS]
        # one ratio per direction only if every measured width agrees (else nothing is quoted)
        self.read_ratio = r[0] if all(v is not None and abs(v - r[0]) < 0.01 for v in r) else None
        self.write_ratio = w[0] if all(v is not None and abs(v - w[0]) < 0.01 for v in w) else None

    @property
    def ok(self):
        return self.read_ratio is not None and self.write_ratio is not None

    def bytes_read(self, fetch_kib):
        return fetch_kib * 1024.0 / self.read_ratio

    def bytes_written(self, write_kib):
        return write_kib * 1024.0 / self.write_ratio

    def describe(self):
        return (f"{self.path}: FETCH_SIZE = {self.read_ratio:.3f} x bytes for coalesced reads of 4 / 8 / 16 B per lane and global_load-to-LDS, "
                f"WRITE_SIZE = {self.write_ratio:.3f} x bytes for coalesced writes (kernels moving a known 2 GiB); bytes = KiB x 1024 / ratio")


class ValuMix:
    """profiles/valu_mix_sketch.json (tools/valu_mix.py): average issue cost of the sketch kernel's VALU instruction mix, derived
    from its disassembly; stale() like PmcFile's."""
    def __init__(self, path="profiles/valu_mix_sketch.json"):
        import json
        self.path = path
        full = path if os.path.isabs(path) else os.path.join(ROOT, path)
        self.doc = json.load(open(full)) if os.path.exists(full) else None

    def stale(self):
        if self.doc is None:
            return f"{self.path} is absent (python tools/valu_mix.py)"
        now = source_hashes()
        changed = [f for f, h in self.doc.get("sources", {}).items() if now.get(f) != h]
        return f"{', '.join(changed)} changed since {self.path} was made (python tools/valu_mix.py)" if changed else None


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "header":
        print(header_line())
    else:
        p = PmcFile(sys.argv[1])
        print(p.sources)
        for r in p.rows:
            print(r)
