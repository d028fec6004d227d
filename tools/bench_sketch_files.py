#!/usr/bin/env python3
"""`sketch` over many gzipped genomes: files/s and Gbase/s of smgpu_sketch_files for 1 and N worker pipelines.
   python tools/bench_sketch_files.py [copies] [threads]     (GPU box; copies of the E. coli K-12 fixture in /tmp)"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from sourmash_amd.sketch import sketch_files  # noqa: E402


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    src = os.path.join(ROOT, "tests", "golden", "ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
    d = "/tmp/many_genomes"
    os.makedirs(d, exist_ok=True)
    paths = []
    for i in range(copies):
        p = os.path.join(d, f"g{i}.fna.gz")
        if not os.path.exists(p):
            shutil.copyfile(src, p)
        paths.append(p)
    bases = 4_641_652 * copies
    out = {"files": copies, "bases": bases, "params": "k=21,k=31,k=51,scaled=1000"}
    sketch_files(paths, out["params"], threads=threads)                      # warm: page cache, pinned buffers, code objects
    for t in (1, 4, threads):
        t0 = time.perf_counter()
        sigs = sketch_files(paths, out["params"], threads=t)
        dt = time.perf_counter() - t0
        assert all(s.minhashes()[1].md5sum() == "0a8632c67e6d88f737ddb510bef90337" for s in sigs)
        out[f"threads_{t}"] = {"s": round(dt, 3), "files_per_s": round(copies / dt, 1), "Gbase_per_s": round(bases / dt / 1e9, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
