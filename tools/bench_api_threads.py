"""compare_all_pairs over 10,000 signature OBJECTS and SketchSet(100,000 MinHash objects): wall clock by the number of host threads
that pack pageable buffers into the pinned ring (SMG_XFER_THREADS; csrc/hostxfer.hpp).   python tools/bench_api_threads.py [threads ...]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    import sourmash_amd as sm
    from sourmash_amd import device as smd
    from sourmash_amd._lowlevel import lib
    from sourmash_amd.compare import compare_all_pairs
    from sourmash_amd.index import SketchSet
    from sourmash_amd.synth import synth_sketches, synth_gather_device
    import ctypes as C

    def xfer(reset=False):
        out = (C.c_uint64 * 5)()
        lib.smgpu_xfer_stats(out, reset)
        return {"h2d_ms": round(out[2] / 1e6, 2), "d2h_ms": round(out[3] / 1e6, 2)}
    dev = torch.device("cuda", 0)
    res = {}
    sk = synth_sketches(10_000, seed=1234)
    sigs = []
    for i, a in enumerate(sk):
        mh = sm.MinHash(0, 31, scaled=1000)
        mh.add_many(a)
        sigs.append(sm.SourmashSignature(mh, name="s%d" % i))
    best = None
    for _ in range(4):
        xfer(True)
        t0 = time.perf_counter()
        compare_all_pairs(sigs, True)
        dt = (time.perf_counter() - t0) * 1e3
        if best is None or dt < best[0]:
            best = (round(dt, 2), xfer())
    res["compare_api_10000"] = {"ms": best[0], **best[1]}
    del sigs, sk
    if os.environ.get("API_GATHER") == "1":
        q, gh, goff = synth_gather_device(1_000_000, 100_000, 5000, dev)
        hh, oo = gh.cpu().numpy().view(np.uint64), goff.cpu().numpy()
        del gh, goff
        mhs = []
        for d in range(100_000):
            mh = sm.MinHash(0, 31, scaled=1000)
            mh.add_many(hh[oo[d]:oo[d + 1]])
            mhs.append(mh)
        best = None
        for _ in range(2):
            xfer(True)
            t0 = time.perf_counter()
            ss = SketchSet(mhs)
            dt = (time.perf_counter() - t0) * 1e3
            if best is None or dt < best[0]:
                best = (round(dt, 2), xfer())
            del ss
        res["sketchset_100000_pack_and_upload"] = {"ms": best[0], **best[1]}
    print(json.dumps(res))


if __name__ == "__main__":
    if os.environ.get("API_CHILD") == "1":
        child()
        sys.exit(0)
    out = {}
    for t in (sys.argv[1:] or ["8", "12", "16"]):
        r = subprocess.run([sys.executable, __file__], capture_output=True, text=True, env=dict(os.environ, API_CHILD="1", SMG_XFER_THREADS=t))
        out[t] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
    print(json.dumps(out))
