"""One ksize, a few sketch steps on resident synthetic DNA: the command the profiler runs for per-k counters.
python tools/bench_sketch_one_k.py <k> [bases] [reps]   -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sourmash_amd import device as smd  # noqa: E402

k = int(sys.argv[1])
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 500_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seq = smd.synth_dna(n, seed=42, record_len=10_000_000)
sk = smd.DeviceSketcher(k, 1000)
sk.sketch(seq)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    h = sk.sketch(seq)
torch.cuda.synchronize()
print(json.dumps({"k": k, "bases": n, "reps": reps, "Gbase_per_s": round(n * reps / (time.perf_counter() - t0) / 1e9, 2), "hashes": int(h.numel())}))
