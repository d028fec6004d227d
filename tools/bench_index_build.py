"""One-shot indexed compare at C3 and C4 (cost model + index build + matrix + mirror + Jaccard, data resident in HBM), and the
index build alone, for both builders of the index (SMG_COMPARE_INDEX=sort selects the radix-sort one).
python tools/bench_index_build.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sourmash_amd import device as smd  # noqa: E402
from sourmash_amd.synth import synth_sketches  # noqa: E402


def main():
    import gc
    gc.collect(); gc.freeze(); gc.disable()
    out = {"builder": os.environ.get("SMG_COMPARE_INDEX", "dict")}
    for name, n in (("C3", 1000), ("C4", 10_000)):
        sk = synth_sketches(n, seed=1234, pool_size=50_000, keep_one_in=10, planted=True)
        h, off = smd.pack_csr(sk)
        pairs = n * (n - 1) // 2
        best_auto, best_build, best_matrix = 1e9, 1e9, 1e9
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c, j = smd.compare_rows(h, off, method="auto")
            torch.cuda.synchronize()
            best_auto = min(best_auto, time.perf_counter() - t0)
            del c, j
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx = smd.BitIndex.build(h, off)
            torch.cuda.synchronize()
            best_build = min(best_build, time.perf_counter() - t0)
            t0 = time.perf_counter()
            c, _ = smd.compare_rows(h, off, index=idx, want_jaccard=False)
            torch.cuda.synchronize()
            best_matrix = min(best_matrix, time.perf_counter() - t0)
            del idx, c
        out[name] = {"pairs": pairs, "auto_ms": round(best_auto * 1e3, 3), "auto_pairs_per_s": round(pairs / best_auto, 1),
                     "index_build_ms": round(best_build * 1e3, 3), "matrix_triangle_and_mirror_ms": round(best_matrix * 1e3, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
