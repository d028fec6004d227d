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


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "header":
        print(header_line())
    else:
        p = PmcFile(sys.argv[1])
        print(p.sources)
        for r in p.rows:
            print(r)
