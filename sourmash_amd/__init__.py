"""sourmash_amd -- MI355X-native FracMinHash engine behind sourmash's API.

Drop-in for the `sourmash sketch` -> `compare` / `search` / `gather` hot path
(SURVEY.md section 8): the Python classes keep the reference's names and
semantics (MinHash, SourmashSignature, compare_all_pairs, CounterGather ...) and
call through a C-ABI (include/sourmash_amd.h) into hand-written HIP kernels for
gfx950.  There is no CPU fallback for k-mer hashing or sketch intersection.
"""
from ._lowlevel import lib as _lib  # noqa: F401  (parses the header; the .so loads on first call)

VERSION = "0.1.0"

from .minhash import MinHash, FrozenMinHash, hash_murmur, get_minhash_default_seed, get_minhash_max_hash  # noqa: E402
from .signature import (SourmashSignature, FrozenSourmashSignature, load_signatures_from_json,  # noqa: E402
                        load_one_signature_from_json, save_signatures_to_json)

DEFAULT_SEED = get_minhash_default_seed()
MAX_HASH = get_minhash_max_hash()


def gpu_available():
    "True if the HIP kernels can run in this process."
    return bool(_lib.smgpu_available())


__all__ = ["MinHash", "FrozenMinHash", "SourmashSignature", "FrozenSourmashSignature", "hash_murmur",
           "load_signatures_from_json", "load_one_signature_from_json", "save_signatures_to_json",
           "gpu_available", "DEFAULT_SEED", "MAX_HASH", "VERSION"]
