set -e
cd $GRAFT_REPO_ROOT
python tools/bench_ingest.py 8e9 2>&1 | tail -4
for t in 2 6; do for c in 33554432; do
echo "threads=$t chunk=$c"; SMG_INGEST_TRACE=1 SMG_INGEST_THREADS=$t SMG_INGEST_CHUNK=$c python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from sourmash_amd.sketch import sketch_file
p = "/tmp/synth.fa"
sketch_file(p, "k=31,scaled=1000")
t0 = time.perf_counter(); sig, = sketch_file(p, "k=31,scaled=1000"); dt = time.perf_counter() - t0
print(f"  {dt:.3f} s  {os.path.getsize(p) / dt / 1e9:.2f} GB/s file bytes")
PY
done; done
