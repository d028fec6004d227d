import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from sourmash_amd import device as smd
from sourmash_amd.synth import synth_sketches
sk = synth_sketches(1000, seed=1234)
h, off = smd.pack_csr(sk)
idx = None
for rep in range(4):
    del idx
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = smd.BitIndex.build(h, off)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    c, j = smd.compare_rows(h, off, index=idx)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"build {1e3*(t1-t0):.3f} ms  matrix+jaccard {1e3*(t2-t1):.3f} ms  stats {idx.stats}")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ca, ja = smd.compare_rows(h, off, method="auto")
    torch.cuda.synchronize(); print(f"one-shot auto (build + matrix + jaccard) {1e3*(time.perf_counter()-t0):.3f} ms", bool((ca == c).all()))
cm, jm = smd.compare_rows(h, off)
torch.cuda.synchronize(); t0 = time.perf_counter()
cm, jm = smd.compare_rows(h, off, common=cm, jaccard=jm)
torch.cuda.synchronize(); print(f"merge {1e3*(time.perf_counter()-t0):.3f} ms", bool((c==cm).all()))
