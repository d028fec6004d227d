#!/usr/bin/env python3
"""Sketch throughput by ksize on resident synthetic DNA (kernel + sort + unique, scaled = 1000): every k from 1 to 88 runs an
instantiation of the register-window kernel, longer k-mers the run-time-k kernel of sketch_words.hip; SMG_SKETCH_GENERIC=1 forces
the byte loop (run in a subprocess, the switch is read once).
python tools/bench_sketch_k.py [long | cross]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ks, n):
    import torch
    from sourmash_amd import device as smd
    seq = smd.synth_dna(n, seed=42, record_len=10_000_000)
    out = {}
    for k in ks:
        sk = smd.DeviceSketcher(k, 1000)
        sk.sketch(seq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            h = sk.sketch(seq)
        torch.cuda.synchronize()
        out[k] = {"Gbase_per_s": round(n * reps / (time.perf_counter() - t0) / 1e9, 2), "hashes": int(h.numel())}
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(json.dumps(run([int(x) for x in sys.argv[3:]], int(sys.argv[2]))))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "long":
        # k above the register-window kernel's 128: the run-time-k kernel of sketch_words.hip, and the byte loop it replaced
        n = 500_000_000
        words = run([88, 89, 96, 112, 128, 129, 144, 160, 200, 256, 300, 1000], n)
        child = subprocess.run([sys.executable, __file__, "child", str(n // 10), "129", "200", "256"], capture_output=True, text=True,
                               env=dict(os.environ, SMG_SKETCH_GENERIC="1"))
        generic = json.loads(child.stdout.strip().splitlines()[-1]) if child.returncode == 0 else {"error": child.stderr[-500:]}
        print(json.dumps({"bases": n, "words_kernel": words, "byte_wise_kernel_forced": generic}))
        sys.exit(0)
    n = 2_000_000_000
    fast = run([15, 21, 25, 27, 31, 33, 41, 51, 63, 64, 65, 80, 88, 89, 96, 112, 127, 128], n)
    child = subprocess.run([sys.executable, __file__, "child", str(n // 10), "25", "33", "65", "127"], capture_output=True, text=True,
                           env=dict(os.environ, SMG_SKETCH_GENERIC="1"))
    generic = json.loads(child.stdout.strip().splitlines()[-1]) if child.returncode == 0 else {"error": child.stderr[-500:]}
    print(json.dumps({"bases": n, "register_window_kernel": fast, "byte_wise_kernel_forced": generic,
                      "k25_over_k31": round(fast[25]["Gbase_per_s"] / fast[31]["Gbase_per_s"], 3)}))
