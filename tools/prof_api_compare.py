import sys, time, cProfile, pstats, io
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import sourmash_amd as sm
from sourmash_amd.compare import compare_all_pairs
from sourmash_amd.synth import synth_sketches
sk = synth_sketches(10_000, seed=1234)
sigs = []
for i, a in enumerate(sk):
    mh = sm.MinHash(0, 31, scaled=1000); mh.add_many(a); sigs.append(sm.SourmashSignature(mh, name="s%d" % i))
for _ in range(2): compare_all_pairs(sigs, True)
t0 = time.perf_counter(); compare_all_pairs(sigs, True); print("wall ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); compare_all_pairs(sigs, True); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3800])
