#!/usr/bin/env python3
"""The .gz ingest by stage (GPU box): one 400 MB FASTA member at zlib level 1 (bench.py's ingest_gz file) and level 6 --
device inflater alone (stage milliseconds from smgpu_gunzip_files) and `sketch` end to end -- and 256 copies of the E. coli
fixture through smgpu_sketch_files.   python tools/bench_gunzip.py [MB] [copies]"""
import json
import os
import shutil
import sys
import tempfile
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from sourmash_amd import device as smd  # noqa: E402
from sourmash_amd.sketch import sketch_file, sketch_files, gunzip_files, gunzip_counters  # noqa: E402


def synth_fasta(path, n, rec=10_000_000):
    seq = smd.synth_dna(n + n // rec, seed=42, record_len=rec, device="cuda").cpu().numpy()
    bases = 0
    with open(path, "wb") as fh:
        for i, r in enumerate(bytes(seq).split(b"\n")):
            a = np.frombuffer(r, dtype=np.uint8)
            bases += len(a)
            fh.write(b">synth_%d\n" % i)
            full = (len(a) // 80) * 80
            if full:
                fh.write(np.concatenate([a[:full].reshape(-1, 80), np.full((full // 80, 1), 10, dtype=np.uint8)], axis=1).tobytes())
            if len(a) > full:
                fh.write(a[full:].tobytes() + b"\n")
    return bases


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    copies = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    tmp = tempfile.mkdtemp(prefix="smg_gz_")
    out = {}
    try:
        plain = os.path.join(tmp, "p.fa")
        bases = synth_fasta(plain, mb * 1_000_000)
        data = open(plain, "rb").read()
        want_md5 = sketch_file(plain, "k=31,scaled=1000")[0].md5sum()
        for level in (1, 6):
            gzp = os.path.join(tmp, f"p{level}.fa.gz")
            co = zlib.compressobj(level, zlib.DEFLATED, 31)
            with open(gzp, "wb") as fo:
                for i in range(0, len(data), 16 << 20):
                    fo.write(co.compress(data[i:i + (16 << 20)]))
                fo.write(co.flush())
            row = {"gz_bytes": os.path.getsize(gzp), "inflated_bytes": len(data), "bases": bases}
            gunzip_files([gzp])
            t0 = time.perf_counter()
            (got,), stats = gunzip_files([gzp])
            row["gunzip_to_host_s"] = round(time.perf_counter() - t0, 3)
            row["identical_to_the_plain_file"] = bool(got == data)
            row["stages"] = {k: round(v, 3) for k, v in stats.items()}
            row["device_GB_per_s_out"] = round(len(data) / (stats["device_total_ms"] * 1e-3) / 1e9, 2)
            del got
            sketch_file(gzp, "k=31,scaled=1000")
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                sig, = sketch_file(gzp, "k=31,scaled=1000")
                ts.append(time.perf_counter() - t0)
            row["sketch_s"] = [round(t, 4) for t in ts]
            row["sketch_Gbase_per_s"] = round(bases / min(ts) / 1e9, 2)
            row["same_sketch_as_the_plain_file"] = bool(sig.md5sum() == want_md5)
            out[f"member_{mb}MB_level{level}"] = row
        del data
        src = os.path.join(ROOT, "tests", "golden", "ecoli", "GCF_000005845.2_ASM584v2_genomic.fna.gz")
        paths = []
        for i in range(copies):
            p = os.path.join(tmp, f"g{i}.fna.gz")
            shutil.copyfile(src, p)
            paths.append(p)
        gb = 4_641_652 * copies
        sketch_files(paths[:16], "k=21,k=31,k=51,scaled=1000", threads=16)
        rows = {}
        for t in (1, 16, 32):
            t0 = time.perf_counter()
            sigs = sketch_files(paths, "k=21,k=31,k=51,scaled=1000", threads=t)
            dt = time.perf_counter() - t0
            ok = all(s.minhashes()[1].md5sum() == "0a8632c67e6d88f737ddb510bef90337" for s in sigs)
            rows[f"threads_{t}"] = {"s": round(dt, 3), "files_per_s": round(copies / dt, 1), "Gbase_per_s": round(gb / dt / 1e9, 3), "golden_md5": ok}
        out[f"ecoli_x{copies}"] = rows
        out["counters_device_host"] = gunzip_counters()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
