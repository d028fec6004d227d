#!/usr/bin/env python3
"""Host-side probe of the GPU box: cores this process may use, cgroup CPU quota, CPU model, and how the CPU oracle
scales with OpenMP threads (input of bench.py's cpu_baseline legs).  Writes one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (lives under tests/: the oracle is test infrastructure)


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def main():
    out = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
           "cgroup_cpu_max": read("/sys/fs/cgroup/cpu.max"),
           "cfs_quota": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cfs_period": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
           "loadavg": read("/proc/loadavg")}
    model = [ln.split(":", 1)[1].strip() for ln in (read("/proc/cpuinfo") or "").splitlines() if ln.startswith("model name")]
    out["cpu_model"] = model[0] if model else None
    seq = oracle.synth_dna(0, 64_000_000, seed=42, record_len=10_000_000)
    rates = {}
    for t in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if t > (os.cpu_count() or 1):
            break
        n = min(len(seq), 4_000_000 * t)
        t0 = time.perf_counter()
        oracle.sketch_dna_bulk(seq[:n], 31, scaled=1000, nthreads=t)
        dt = time.perf_counter() - t0
        rates[t] = round(n / dt / 1e6, 2)
    out["oracle_sketch_Mbase_per_s_by_threads"] = rates
    print(json.dumps(out))


if __name__ == "__main__":
    main()
