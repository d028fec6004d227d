#!/usr/bin/env python3
"""What does a VALU instruction of sketch_dna_kernel<31,16,false> cost on average?  

bench.py turns the counter SQ_INSTS_VALU into "fraction of the SIMDs' issue cycles" with the average cost of the kernel's
instruction MIX.  Round 1-3 carried that average as a constant (3.76) from a one-off count.  This tool derives it from the code:

  1. compiles csrc/sketch.hip for gfx950 to assembly (hipcc --cuda-device-only -S; no GPU needed),
  2. takes the kernel's main loop (one tile of 256 lanes x 16 positions per trip) and splits it into the HOT path and the
     blocks that a wave-level branch (s_cbranch_vccz / vccnz / scc0 / scc1 over a forward region: "some lane of the wave has a
     bad byte / a tie / a hash that can still pass") skips -- bad bytes and canonical ties do not occur on random DNA, the hash is
     finished in 1 wave-step of 16 at scaled = 1000 (kmer_core.hpp),
  3. histograms the VALU opcodes of each part and prices every opcode with the measured issue cost of its class
     (profiles/r01_ubench_valu.txt, waves/SIMD = 4 rows; opcodes that were not micro-benchmarked are listed and priced as
     half-rate, the class everything but and/or/xor/add/sub/lshr/mov/bitop3 fell into),
  4. writes profiles/valu_mix_sketch.json with the source hashes of the kernel's files: bench.py refuses it when they differ.

The hot-path instruction count per k-mer is printed next to the counter value (116.5 per k-mer, profiles/r03_pmc.txt) as a check
that the split is the right one.

usage: python tools/valu_mix.py [--keep-asm PATH]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
from pmcfile import source_hashes  # noqa: E402

CSRC = os.path.join(ROOT, "sourmash_amd", "csrc")
KERNEL = "sketch_dna_kernelILi31ELi16ELb0E"
SOURCES = ["sketch.hip", "sketch_kernel.hpp", "kmer_core.hpp", "murmur3.hpp"]
FINISH_PROB = 1.0 / 16.0      # wave-steps whose hash is finished at scaled = 1000 (early reject on the top dword)
KEEP_PROB = 1.0 - (1.0 - 1.0 / 1000.0) ** 64   # wave-steps in which some lane keeps a hash at scaled = 1000 (the LDS append)

# opcode (without encoding suffix) -> micro-benchmark kernel of profiles/r01_ubench_valu.txt
UBENCH = {
    "v_and_b32": "k_and_e32", "v_or_b32": "k_or_e32", "v_xor_b32": "k_xor", "v_lshlrev_b32": "k_lshl_e32", "v_lshrrev_b32": "k_lshr_e32",
    "v_sub_u32": "k_sub_e32", "v_subrev_u32": "k_sub_e32", "v_mov_b32": "k_mov_e32", "v_add_u32": "k_add", "v_add_co_u32": "k_addco_e32",
    "v_addc_co_u32": "k_addc_e32", "v_subb_co_u32": "k_addc_e32", "v_sub_co_u32": "k_addco_e32", "v_cndmask_b32": "k_cndmask",
    "v_and_or_b32": "k_and_or", "v_lshl_or_b32": "k_lshl_or", "v_or3_b32": "k_or3", "v_bfe_u32": "k_bfe", "v_mul_u32_u24": "k_mul_u24",
    "v_mad_u32_u24": "k_mad_u24", "v_bitop3_b32": "k_bitop3_xor3", "v_add3_u32": "k_add3", "v_lshl_add_u32": "k_lshladd", "v_perm_b32": "k_perm",
    "v_alignbit_b32": "k_alignbit", "v_alignbyte_b32": "k_alignbit", "v_mul_lo_u32": "k_mul_lo", "v_mul_hi_u32": "k_mul_hi",
    "v_mad_u64_u32": "k_mad_u64_u32", "v_lshl_add_u64": "k_lshl_add_u64", "v_lshlrev_b64": "k_lshlrev_b64", "v_lshrrev_b64": "k_lshrrev_b64",
    "v_xor3_b32": "k_bitop3_xor3", "v_mov_b64": "k_mov_e32",
}


def ubench_costs():
    costs = {}
    with open(os.path.join(ROOT, "profiles", "r01_ubench_valu.txt")) as f:
        for line in f:
            m = re.match(r"(k_\w+)\s+waves/SIMD=4\s+[\d.]+ ms\s+([\d.]+) cycles", line)
            if m:
                costs[m.group(1)] = float(m.group(2))
    return costs


def opcode_cost(op, costs):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_cmp") or base.startswith("v_cmpx"):
        return costs["k_cmp_lt_u64"] if base.endswith("64") else costs["k_cmp_lt_u32_e32"], True
    k = UBENCH.get(base)
    if k and k in costs:
        return costs[k], True
    return costs["k_perm"], False            # not micro-benchmarked: priced as half-rate


def main():
    keep = sys.argv[sys.argv.index("--keep-asm") + 1] if "--keep-asm" in sys.argv else None
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        asm = keep or os.path.join(td, "sketch.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
                               os.path.join(CSRC, "sketch.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN3smg") and KERNEL in ln and ln.rstrip().endswith(tuple(": ;")) or (KERNEL in ln and re.match(r"^_ZN3smg\S+:", ln)))
    body = []
    for ln in lines[start + 1:]:
        body.append(ln)
        if "s_endpgm" in ln:
            break
    # instruction stream with labels
    items = []                                  # (kind, text): kind = "label" | "inst"
    for ln in body:
        t = ln.split(";")[0].strip()
        if not t:
            continue
        if re.match(r"^\.LBB\d+_\d+:", t):
            items.append(("label", t[:-1]))
        elif not t.startswith("."):
            items.append(("inst", t))
    label_at = {t: i for i, (k, t) in enumerate(items) if k == "label"}
    # the main loop: the backward branch that spans the most instructions
    best = None
    for i, (k, t) in enumerate(items):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) if k == "inst" else None
        if not m:
            m = re.match(r"s_branch\s+(\.LBB\d+_\d+)", t) if k == "inst" else None
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            span = i - label_at[m.group(1)]
            if best is None or span > best[1] - best[0]:
                best = (label_at[m.group(1)], i)
    lo, hi = best
    # forward regions skipped by a wave-level branch (class 1), or entered only by the lanes of an exec mask that may be empty
    # (s_cbranch_execz: class 2 -- appending a kept hash, 1 position in 1,000; flushing the workgroup's buffer)
    cold = [0] * len(items)
    for i in range(lo, hi):
        k, t = items[i]
        m = re.match(r"s_cbranch_(vccz|vccnz|scc0|scc1|execz)\s+(\.LBB\d+_\d+)", t) if k == "inst" else None
        if m and m.group(2) in label_at and i < label_at[m.group(2)] <= hi:
            cls = 2 if m.group(1) == "execz" else 1
            for j in range(i + 1, label_at[m.group(2)]):
                cold[j] = cold[j] or cls
    costs = ubench_costs()
    parts = {"hot": {}, "wave_conditional": {}, "lane_conditional": {}}
    for i in range(lo, hi + 1):
        k, t = items[i]
        if k != "inst":
            continue
        op = t.split()[0]
        if not op.startswith("v_") or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_nop")):
            continue
        d = parts[("hot", "wave_conditional", "lane_conditional")[cold[i]]]
        d[op] = d.get(op, 0) + 1
    positions = 16
    out = {"kernel": "sketch_dna_kernel<31, 16, false>", "sources": {f: source_hashes()[f] for f in SOURCES},
           "method": "static opcode histogram of the main loop (one trip = 16 positions per lane), hot path and wave-conditional blocks "
                     "apart; costs per opcode class from profiles/r01_ubench_valu.txt (waves/SIMD = 4)",
           "finish_probability": FINISH_PROB, "keep_probability": round(KEEP_PROB, 4), "unmeasured_opcodes": []}
    tot_n = tot_c = 0.0
    for name, hist in parts.items():
        weight = {"hot": 1.0, "wave_conditional": FINISH_PROB, "lane_conditional": KEEP_PROB}[name]
        n = sum(hist.values())
        cyc = 0.0
        rows = {}
        for op, c in sorted(hist.items(), key=lambda kv: -kv[1]):
            cost, known = opcode_cost(op, costs)
            if not known and op not in out["unmeasured_opcodes"]:
                out["unmeasured_opcodes"].append(op)
            cyc += c * cost
            rows[op] = {"count": c, "cycles_each": cost}
        out[name] = {"valu_insts_per_trip": n, "valu_insts_per_kmer": round(n / positions, 2), "cycles_per_inst": round(cyc / n, 3) if n else None,
                     "opcodes": rows}
        tot_n += weight * n
        tot_c += weight * cyc
    out["expected_valu_insts_per_kmer"] = round(tot_n / positions, 2)
    out["mix_cycles_per_valu_inst"] = round(tot_c / tot_n, 3)
    half = sum(c["count"] for c in out["hot"]["opcodes"].values() if c["cycles_each"] > 3.5)
    out["hot_half_rate_fraction"] = round(half / out["hot"]["valu_insts_per_trip"], 3)
    path = os.path.join(ROOT, "profiles", "valu_mix_sketch.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{path}: hot path {out['hot']['valu_insts_per_kmer']} VALU / k-mer at {out['hot']['cycles_per_inst']} cycles, wave-conditional "
          f"{out['wave_conditional']['valu_insts_per_kmer']} / k-mer x {FINISH_PROB:.4f}, lane-conditional {out['lane_conditional']['valu_insts_per_kmer']} / k-mer x {KEEP_PROB:.4f}; expected {out['expected_valu_insts_per_kmer']} / k-mer "
          f"(counter: 116.5), mix {out['mix_cycles_per_valu_inst']} cycles per VALU instruction; unmeasured opcodes priced half-rate: "
          f"{out['unmeasured_opcodes']}")


if __name__ == "__main__":
    main()
