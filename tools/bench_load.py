#!/usr/bin/env python3
"""Throughput of the bulk signature loader: N synthetic signatures of ~5,000 hashes
written as a sourmash-style zip (stored signatures/<md5>.sig.gz members + SOURMASH-MANIFEST.csv), then loaded
  (a) by the native multi-threaded loader into a host CSR (and, with a GPU, into HBM),
  (b) one signature at a time through the object API (what per-sketch Python objects cost), on a sample.

    python tools/bench_load.py [--n 10000] [--threads 0]
"""
import argparse
import gzip
import hashlib
import io
import json
import os
import sys
import tempfile
import time
import zipfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sourmash_amd.synth import splitmix64, MAX_HASH_1000  # noqa: E402


def write_zip(path, n, size=5000):
    t0 = time.perf_counter()
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as zf:
        man = io.StringIO()
        man.write("# SOURMASH-MANIFEST-VERSION: 1.0\n")
        man.write("internal_location,md5,md5short,ksize,moltype,num,scaled,n_hashes,with_abundance,name,filename\r\n")
        for i in range(n):
            mins = np.unique(splitmix64((np.uint64(i) << np.uint64(32)) + np.arange(size, dtype=np.uint64)) % np.uint64(MAX_HASH_1000))
            text_mins = ",".join(map(str, mins.tolist()))
            md5 = hashlib.md5(("31" + text_mins.replace(",", "")).encode()).hexdigest()
            doc = ('[{"class":"sourmash_signature","email":"","hash_function":"0.murmur64","filename":"g%d.fa","name":"genome %d",'
                   '"license":"CC0","signatures":[{"num":0,"ksize":31,"seed":42,"max_hash":%d,"mins":[%s],"md5sum":"%s",'
                   '"molecule":"dna"}],"version":0.4}]' % (i, i, MAX_HASH_1000, text_mins, md5))
            loc = f"signatures/{md5}.sig.gz"
            zf.writestr(loc, gzip.compress(doc.encode(), compresslevel=1))
            man.write(f"{loc},{md5},{md5[:8]},31,DNA,0,1000,{len(mins)},0,genome {i},g{i}.fa\r\n")
        zf.writestr("SOURMASH-MANIFEST.csv", man.getvalue(), compress_type=zipfile.ZIP_DEFLATED)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--sample", type=int, default=300, help="signatures loaded through the object API for comparison")
    args = ap.parse_args()
    from sourmash_amd import index
    import sourmash_amd as sm
    out = {"n": args.n, "host_cores": os.cpu_count()}
    with tempfile.TemporaryDirectory() as tmp:
        zpath = os.path.join(tmp, "coll.zip")
        out["write_s"] = round(write_zip(zpath, args.n), 2)
        out["zip_bytes"] = os.path.getsize(zpath)
        for threads in ([1, args.threads] if args.threads != 1 else [1]):
            t0 = time.perf_counter()
            col = index.Collection(zpath, ksize=31, moltype="DNA", threads=threads)
            dt = time.perf_counter() - t0
            out[f"native_threads_{threads or 'all'}"] = {
                "s": round(dt, 3), "sigs_per_s": round(len(col) / dt, 1), "hashes_per_s": round(col.total_hashes / dt, 1),
                "zip_MBps": round(out["zip_bytes"] / dt / 1e6, 1)}
        out["rows"], out["total_hashes"] = len(col), int(col.total_hashes)
        if sm.gpu_available():
            t0 = time.perf_counter()
            db = index.SketchSet.load(zpath, ksize=31, moltype="DNA", threads=args.threads)
            out["to_hbm_s"] = round(time.perf_counter() - t0, 3)
            assert len(db) == args.n
        if True:
            # per-object route on a sample: inflate + parse + one MinHash/Signature object each
            with zipfile.ZipFile(zpath) as zf:
                names = [n for n in zf.namelist() if n.endswith(".sig.gz")][:args.sample]
                t0 = time.perf_counter()
                objs = [next(sm.load_signatures_from_json(zf.read(n))) for n in names]
                dt = time.perf_counter() - t0
            out["object_route"] = {"sample": len(objs), "sigs_per_s": round(len(objs) / dt, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
