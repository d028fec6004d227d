#!/usr/bin/env python3
"""SketchSet.load of a sourmash-style zip (stored .sig.gz members + manifest) of N sketches of ~5,000 hashes: the device loader
(csrc/sigload.hpp) and, in a child process with SMG_SIGLOAD_DEVICE=0, the host loader of the same library.
   python tools/bench_sigload.py [N ...]        (GPU box; default 10000 100000)"""
import gzip
import hashlib
import io
import json
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile
import time
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def make_doc(i):
    from sourmash_amd.synth import splitmix64, MAX_HASH_1000
    mins = np.unique(splitmix64((np.uint64(i) << np.uint64(32)) + np.arange(5000, dtype=np.uint64)) % np.uint64(MAX_HASH_1000))
    text = ",".join(map(str, mins.tolist()))
    md5 = hashlib.md5(("31" + text.replace(",", "")).encode()).hexdigest()
    doc = ('[{"class":"sourmash_signature","email":"","hash_function":"0.murmur64","filename":"g%d.fa","name":"genome %d",'
           '"license":"CC0","signatures":[{"num":0,"ksize":31,"seed":42,"max_hash":%d,"mins":[%s],"md5sum":"%s",'
           '"molecule":"dna"}],"version":0.4}]' % (i, i, MAX_HASH_1000, text, md5))
    return md5, len(mins), gzip.compress(doc.encode(), compresslevel=1)


def write_zip(path, n):
    with mp.Pool(min(16, os.cpu_count() or 1)) as pool, zipfile.ZipFile(path, "w", zipfile.ZIP_STORED, allowZip64=True) as zf:
        man = io.StringIO()
        man.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
        man.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n")
        for i, (md5, nh, blob) in enumerate(pool.imap(make_doc, range(n), chunksize=64)):
            loc = f"signatures/{md5}.sig.gz"
            zf.writestr(loc, blob)
            man.write(f"{loc},{md5},{md5[:8]},31,DNA,0,1000,{nh},0,genome {i},g{i}.fa\r\n")
        zf.writestr("SOURMASH-MANIFEST.csv", man.getvalue(), compress_type=zipfile.ZIP_DEFLATED)


def time_load(path):
    import torch  # noqa: F401
    from sourmash_amd.index import SketchSet
    SketchSet.load(path, ksize=31, moltype="DNA")
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        db = SketchSet.load(path, ksize=31, moltype="DNA")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"seconds": round(best, 4), "signatures": len(db), "signatures_per_s": round(len(db) / best, 1), "total_hashes": int(db.total_hashes),
            "checksum": int(np.bitwise_xor.reduce(db.sizes.astype(np.uint64) * np.arange(1, len(db) + 1, dtype=np.uint64)))}


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        print(json.dumps(time_load(sys.argv[2])))
        return
    sizes = [int(a) for a in sys.argv[1:]] or [10_000, 100_000]
    out = {}
    tmp = tempfile.mkdtemp(prefix="smg_sigload_")
    try:
        for n in sizes:
            z = os.path.join(tmp, f"coll{n}.zip")
            t0 = time.perf_counter()
            write_zip(z, n)
            row = {"zip_bytes": os.path.getsize(z), "written_in_s": round(time.perf_counter() - t0, 1)}
            row["device"] = time_load(z)
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", z], env=dict(os.environ, SMG_SIGLOAD_DEVICE="0"),
                                   capture_output=True, text=True)
            row["host"] = json.loads(child.stdout.strip().splitlines()[-1]) if child.returncode == 0 else {"error": child.stderr[-500:]}
            row["same_rows"] = row["host"].get("checksum") == row["device"]["checksum"] and row["host"].get("total_hashes") == row["device"]["total_hashes"]
            out[f"sigload_{n}"] = row
            os.remove(z)
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
