#!/usr/bin/env python3
"""End-to-end file -> sketch throughput of the native ingest path (GPU box; writes to /tmp)."""
import os, sys, time, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sourmash_amd.sketch import sketch_file
from sourmash_amd import device as smd

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
rec = 10_000_000
seq = smd.synth_dna(n + n // rec, seed=42, record_len=rec).cpu().numpy()
path = "/tmp/synth.fa"
t0 = time.perf_counter()
with open(path, "wb") as fh:
    recs = bytes(seq).split(b"\n")
    for i, r in enumerate(recs):
        fh.write(b">synth_%d\n" % i)
        a = np.frombuffer(r, dtype=np.uint8)
        full = (len(a) // 80) * 80
        if full:
            lines = np.concatenate([a[:full].reshape(-1, 80), np.full((full // 80, 1), 10, dtype=np.uint8)], axis=1)
            fh.write(lines.tobytes())
        if len(a) > full:
            fh.write(a[full:].tobytes() + b"\n")
print(f"wrote {os.path.getsize(path) / 1e9:.2f} GB FASTA in {time.perf_counter() - t0:.1f} s")
for p in ("k=31,scaled=1000", "k=21,k=31,k=51,scaled=1000"):
    sketch_file(path, p)                      # warm (page cache, allocations)
    t0 = time.perf_counter()
    sig, = sketch_file(path, p)
    dt = time.perf_counter() - t0
    bases = sum(len(r) for r in recs)
    print(f"{p}: {dt:.2f} s  {bases / dt / 1e9:.2f} Gbase/s end to end  ({len(sig.minhash)} hashes)")

if "--gz" in sys.argv:
    # ONE gzip member of the same FASTA (level 1, as `gzip -1` writes it): many-thread inflate (csrc/pargz.hpp) vs one zlib stream
    import subprocess
    import zlib
    gz = path + ".gz"
    t0 = time.perf_counter()
    co = zlib.compressobj(1, zlib.DEFLATED, 31)
    with open(path, "rb") as fi, open(gz, "wb") as fo:
        while True:
            block = fi.read(16 << 20)
            if not block:
                break
            fo.write(co.compress(block))
        fo.write(co.flush())
    print(f"wrote {os.path.getsize(gz) / 1e9:.2f} GB gzip member in {time.perf_counter() - t0:.1f} s")
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from sourmash_amd.sketch import sketch_file\n"
            "sketch_file(%r, 'k=31,scaled=1000')\n"
            "t0 = time.perf_counter(); sig, = sketch_file(%r, 'k=31,scaled=1000'); dt = time.perf_counter() - t0\n"
            "print(dt, sig.md5sum())\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), gz, gz))
    sig_plain, = sketch_file(path, "k=31,scaled=1000")
    for label, env in (("many threads", {}), ("one zlib stream", {"SMG_GUNZIP_SEQUENTIAL": "1"})):
        out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, **env), text=True).split()
        dt, md5 = float(out[-2]), out[-1]
        print(f"gz, {label}: {dt:.2f} s  {bases / dt / 1e9:.2f} Gbase/s end to end; md5 equals the plain file's: {md5 == sig_plain.md5sum()}")
