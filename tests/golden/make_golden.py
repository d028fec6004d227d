#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference checkout.

The reference (sourmash @ /root/reference) cannot be imported or built in this
image (Rust core, no cargo; no cffi/screed wheels -- SURVEY.md section 8c), so
the golden vectors are the reference's OWN committed fixtures: input FASTA files
together with the signatures the reference produced from them, and the
signature collections its tests assert exact results on.  This script copies
those data files verbatim (no source code) and records where each came from in
MANIFEST.json.  Run from the repo root inside the build container:

    python tests/golden/make_golden.py

/root/reference does not exist on the GPU box; tests only read tests/golden/.
"""
import hashlib
import json
import os
import shutil
import sys

REF = os.environ.get("SOURMASH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

# (source path under REF, destination under tests/golden, why / which reference test pins it)
FILES = [
    ("data/GCF_000005845.2_ASM584v2_genomic.fna.gz", "ecoli/GCF_000005845.2_ASM584v2_genomic.fna.gz",
     "BASELINE config C1 input (E. coli K-12)"),
    ("tests/test-data/GCF_000005845.2_ASM584v2_genomic.fna.gz.sig", "ecoli/GCF_000005845.2_ASM584v2_genomic.fna.gz.sig",
     "reference sketch of the above: k=21/31/51 scaled=1000 (k=31: 4476 hashes, md5 0a8632c6...)"),
    ("data/GCF_000006945.1_ASM694v1_genomic.fna.gz", "scaled100/GCF_000006945.1_ASM694v1_genomic.fna.gz",
     "second genome, input"),
    ("tests/test-data/scaled100/GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz",
     "scaled100/GCF_000006945.1_ASM694v1_genomic.fna.gz.sig.gz", "reference sketch k=21 scaled=100 (48504 hashes)"),
    ("tests/test-data/scaled100/GCF_000005845.2_ASM584v2_genomic.fna.gz.sig.gz",
     "scaled100/GCF_000005845.2_ASM584v2_genomic.fna.gz.sig.gz",
     "reference sketch of the E. coli genome k=21 scaled=100 (45577 hashes); tests/test_jaccard.py:207-232"),
    ("tests/test-data/genome-s10.fa.gz", "num/genome-s10.fa.gz", "input for the num (bottom-k) sketch"),
    ("tests/test-data/genome-s10.fa.gz.sig", "num/genome-s10.fa.gz.sig", "reference sketch k=21/30 num=500 (+protein)"),
    ("tests/test-data/ecoli.genes.fna", "genes/ecoli.genes.fna", "tests/test_sourmash_compute.py:858-897 input"),
    ("tests/test-data/benchmark.dna.sig", "genes/benchmark.dna.sig",
     "independent mmh3 restatement output (utils/compute-dna-mh-another-way.py)"),
    ("tests/test-data/47.fa.sig", "pairs/47.fa.sig", "tests/test_jaccard.py:175-232"),
    ("tests/test-data/63.fa.sig", "pairs/63.fa.sig", "tests/test_jaccard.py:175-232"),
    ("tests/test-data/track_abund/47.fa.sig", "pairs/track_abund_47.fa.sig", "abundance sketches (angular similarity)"),
    ("tests/test-data/track_abund/63.fa.sig", "pairs/track_abund_63.fa.sig", "abundance sketches (angular similarity)"),
]
for name in ["SRR2060939_1", "SRR2060939_2", "SRR2241509_1", "SRR2255622_1", "SRR453566_1", "SRR453569_1", "SRR453570_1"]:
    FILES.append((f"tests/test-data/demo/{name}.sig", f"demo/{name}.sig", "tests/test_compare.py:48-63 7x7 matrix"))
for name in ["GCF_000006945.2_ASM694v2", "GCF_000007545.1_ASM754v1", "GCF_000008105.1_ASM810v1",
             "GCF_000008545.1_ASM854v1", "GCF_000009085.1_ASM908v1", "GCF_000009505.1_ASM950v1",
             "GCF_000009525.1_ASM952v1", "GCF_000011885.1_ASM1188v1", "GCF_000016045.1_ASM1604v1",
             "GCF_000016785.1_ASM1678v1", "GCF_000018945.1_ASM1894v1", "GCF_000195995.1_ASM19599v1"]:
    FILES.append((f"tests/test-data/gather/{name}_genomic.fna.gz.sig", f"gather/{name}_genomic.fna.gz.sig",
                  "tests/test_index_protocol.py:1057-1097 golden gather"))
FILES.append(("tests/test-data/gather/combined.sig", "gather/combined.sig", "golden gather query"))
for name in ["genome-s10.fa.gz.sig", "genome-s11.fa.gz.sig", "genome-s12.fa.gz.sig", "reads-s10-s11.sig", "reads-s10x10-s11.sig"]:
    FILES.append((f"tests/test-data/gather-abund/{name}", f"gather-abund/{name}",
                  "tests/test_sourmash.py:6386-6600 abundance-weighted gather (p_query / p_match / avg_abund columns)"))
for name, why in (("ecoli.faa", "protein input, tests/test_sourmash_compute.py:811-930"),
                  ("benchmark.input_prot.sig", "independent restatement of protein-input hashing (utils/compute-input-prot-another-way.py)"),
                  ("benchmark.prot.sig", "independent restatement of six-frame translation hashing (utils/compute-prot-mh-another-way.py)")):
    FILES.append((f"tests/test-data/{name}", f"genes/{name}", why))
FILES.append(("tests/test-data/2+63.fa.sig", "pairs/2+63.fa.sig", "tests/test_compare.py:94-196 (ANI matrices of 2 / 2+63 / 47 / 63)"))
FILES.append(("tests/test-data/2.fa.sig", "pairs/2.fa.sig", "tests/test_index_protocol.py:31-700 (the three-signature index cases)"))
FILES.append(("tests/test-data/47+63.fa.sig", "pairs/47+63.fa.sig", "tests/test_search.py:257-590 result-row fixtures"))
FILES.append(("tests/test-data/track_abund/track_abund.zip", "zips/track_abund.zip",
              "a zip as `sourmash sig cat -o x.zip` writes it: stored signatures/<md5>.sig.gz members + SOURMASH-MANIFEST.csv"))
FILES.append(("tests/test-data/prot/all.zip", "zips/all.zip",
              "tests/test_index.py:821-906: deflated members, directories, a non-signature member, 8 manifest rows "
              "(DNA x2, protein/dayhoff/hp x2 each)"))


def reference_distance_utils():
    """the reference's src/sourmash/distance_utils.py imported on its own: it is plain Python + scipy and its one package import
    is `.logging.notify`, so it runs in this container without the Rust core (the rest of the package does not)"""
    import importlib.util
    import types
    pkg = types.ModuleType("sourmash")
    pkg.__path__ = []
    sys.modules["sourmash"] = pkg
    lg = types.ModuleType("sourmash.logging")
    lg.notify = lambda *a, **k: None
    sys.modules["sourmash.logging"] = lg
    spec = importlib.util.spec_from_file_location("sourmash.distance_utils", os.path.join(REF, "src/sourmash/distance_utils.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["sourmash.distance_utils"] = m
    spec.loader.exec_module(m)
    return m


def jaccard_to_distance_vectors():
    """tests/golden/jaccard_to_distance.json: outputs of the REFERENCE's jaccard_to_distance (distance_utils.py:349-407) on 400 seeded
    inputs, floats as hex -- pins the oracle's restatement and the batched ANI matrix of compare (compare.py:14-64 return_ani)."""
    import random
    m = reference_distance_utils()
    random.seed(5)
    cases = []
    for _ in range(400):
        j = random.random() ** random.choice([1, 2, 4])
        k = random.choice([21, 31, 51, 7, 10])
        n = random.choice([100, 5000, 50000, 5_000_000, 12345678])
        try:
            r = m.jaccard_to_distance(j, k, 1000, n_unique_kmers=n)
            cases.append([j.hex(), k, n, [r.dist.hex(), r.jaccard_error.hex(), r.ani is None]])
        except ValueError:
            cases.append([j.hex(), k, n, None])
    doc = {"source": "src/sourmash/distance_utils.py:349-407 jaccard_to_distance(j, ksize, 1000, n_unique_kmers=n): [jaccard hex, "
                     "ksize, n, [dist hex, jaccard_error hex, ani is None] or null where it raises]", "cases": cases}
    with open(os.path.join(HERE, "jaccard_to_distance.json"), "w") as fh:
        json.dump(doc, fh)


def containment_to_distance_vectors():
    """tests/golden/containment_to_distance.json: outputs of the REFERENCE's containment_to_distance (distance_utils.py:258-346,
    point estimate + confidence interval), get_exp_probability_nothing_common (:233-255), set_size_exact_prob (:198-231) and
    set_size_chernoff (:181-196) on seeded inputs, floats as hex -- pins the host float layer of this package
    (sourmash_amd/distance_utils.py) to the reference module itself, beyond the handful of values its tests print."""
    import random
    m = reference_distance_utils()
    random.seed(11)
    cases = []
    for i in range(150):
        c = random.random() ** random.choice([1, 3])
        if i % 25 == 0:
            c = [0.0, 1.0][(i // 25) % 2]
        k = random.choice([21, 31, 51, 7, 10])
        scaled = random.choice([1, 100, 1000, 10000])
        n = random.choice([1000, 50000, 5_000_000])
        try:
            r = m.containment_to_distance(c, k, scaled, n_unique_kmers=n, estimate_ci=True)
            out = [r.dist.hex(), None if r.dist_low is None else float(r.dist_low).hex(), None if r.dist_high is None else float(r.dist_high).hex(),
                   float(r.p_nothing_in_common).hex(), bool(r.p_exceeds_threshold)]
        except ValueError:
            out = None
        cases.append(["c2d", c.hex(), k, scaled, n, out])
    for i in range(60):
        n = random.choice([100, 5000, 123456, 5_000_000])
        scaled = random.choice([1, 10, 100, 1000])
        cases.append(["size", n, scaled, float(m.set_size_exact_prob(n, scaled, relative_error=0.2)).hex(), float(m.set_size_chernoff(n, scaled, relative_error=0.2)).hex()])
    doc = {"source": "src/sourmash/distance_utils.py: containment_to_distance(c, ksize, scaled, n_unique_kmers=n, estimate_ci=True) -> "
                     "[dist, dist_low, dist_high, p_nothing_in_common (hex), p_exceeds_threshold] or null where it raises; "
                     "set_size_exact_prob / set_size_chernoff(n, scaled, relative_error=0.2)", "cases": cases}
    with open(os.path.join(HERE, "containment_to_distance.json"), "w") as fh:
        json.dump(doc, fh)


def main():
    jaccard_to_distance_vectors()
    containment_to_distance_vectors()
    manifest = []
    for src, dst, why in FILES:
        s = os.path.join(REF, src)
        d = os.path.join(HERE, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(s, d)
        with open(d, "rb") as fh:
            sha = hashlib.sha256(fh.read()).hexdigest()
        manifest.append({"file": dst, "reference_path": src, "sha256": sha, "pins": why})
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print(f"copied {len(manifest)} fixtures", file=sys.stderr)


if __name__ == "__main__":
    main()
