"""The overlap pass (search / prefetch: |Q ∩ row| for every row) at config-C5 size on one GPU:
python tools/bench_overlap.py [--ndb 100000] [--reps 5]   (SMG_OVERLAP / SMG_OVERLAP_ROWS select the form)
-> one JSON line: ms per pass (HIP events), algorithmic GB/s, and two checksums of the counts (equal across forms)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sourmash_amd import parallel  # noqa: E402
from sourmash_amd.synth import synth_gather_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=1_000_000)
    ap.add_argument("--ndb", type=int, default=100_000)
    ap.add_argument("--dbsize", type=int, default=5000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    be = parallel.DeviceBackend(dev)
    q, h, off = synth_gather_device(a.nq, a.ndb, a.dbsize, dev)
    cnt = be.zeros((a.ndb,), torch.int64)
    be.overlaps(q, q.numel(), h, off, a.ndb, cnt, 0)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
    for e0, e1 in evs:
        e0.record()
        be.overlaps(q, q.numel(), h, off, a.ndb, cnt, 0)
        e1.record()
    torch.cuda.synchronize()
    ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    db_bytes = int(h.numel()) * 8
    w = torch.arange(1, a.ndb + 1, device=dev, dtype=torch.int64)
    print(json.dumps({"form": os.environ.get("SMG_OVERLAP", "auto"),
                      "rows_env": os.environ.get("SMG_OVERLAP_ROWS"), "ndb": a.ndb, "db_bytes": db_bytes,
                      "ms_min": round(ms[0], 3), "ms_median": round(ms[len(ms) // 2], 3),
                      "GBps_algorithmic": round((db_bytes + 8 * int(q.numel())) / (ms[len(ms) // 2] * 1e-3) / 1e9, 1),
                      "sum": int(cnt.sum().item()), "weighted": int((cnt * w % 1000003).sum().item())}))


if __name__ == "__main__":
    main()
