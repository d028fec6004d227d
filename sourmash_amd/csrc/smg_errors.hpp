// Error model of the C-ABI: numeric codes are ABI (include/sourmash.h:19-53,
// src/core/src/errors.rs:101-143); messages follow src/core/src/errors.rs:5-99.
// Convention (src/core/src/ffi/utils.rs:17-19,58-83,195-207): a fallible entry
// point stores the error in thread-local state and returns an all-zero value;
// the caller polls sourmash_err_get_last_code().
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>

namespace smg {

enum ErrorCode : uint32_t {
    E_NO_ERROR = 0,
    E_PANIC = 1,
    E_INTERNAL = 2,
    E_MSG = 3,
    E_UNKNOWN = 4,
    E_MISMATCH_KSIZES = 101,
    E_MISMATCH_DNA_PROT = 102,
    E_MISMATCH_SCALED = 103,
    E_MISMATCH_SEED = 104,
    E_MISMATCH_SIGNATURE_TYPE = 105,
    E_NON_EMPTY_MINHASH = 106,
    E_MISMATCH_NUM = 107,
    E_NEEDS_ABUNDANCE_TRACKING = 108,
    E_CANNOT_UPSAMPLE_SCALED = 109,
    E_NO_MINHASH_FOUND = 110,
    E_EMPTY_SIGNATURE = 111,
    E_MULTIPLE_SKETCHES_FOUND = 112,
    E_INVALID_DNA = 1101,
    E_INVALID_PROT = 1102,
    E_INVALID_CODON_LENGTH = 1103,
    E_INVALID_HASH_FUNCTION = 1104,
    E_READ_DATA = 1201,
    E_STORAGE = 1202,
    E_HLL_PRECISION_BOUNDS = 1301,
    E_ANI_ESTIMATION = 1401,
    E_IO = 100001,
    E_UTF8 = 100002,
    E_PARSE_INT = 100003,
    E_SERDE = 100004,
    E_NIFFLER = 100005,
    E_CSV = 100006,
    E_ROCKSDB = 100007,
};

struct Error : public std::runtime_error {
    uint32_t code;
    Error(uint32_t c, const std::string& msg) : std::runtime_error(msg), code(c) {}
};

inline Error err_mismatch_ksizes() { return Error(E_MISMATCH_KSIZES, "different ksizes cannot be compared"); }
inline Error err_mismatch_dnaprot() { return Error(E_MISMATCH_DNA_PROT, "DNA/prot minhashes cannot be compared"); }
inline Error err_mismatch_scaled() { return Error(E_MISMATCH_SCALED, "mismatch in scaled; comparison fail"); }
inline Error err_mismatch_seed() { return Error(E_MISMATCH_SEED, "mismatch in seed; comparison fail"); }
inline Error err_needs_abundance() { return Error(E_NEEDS_ABUNDANCE_TRACKING, "sketch needs abundance for this operation"); }
inline Error err_cannot_upsample() { return Error(E_CANNOT_UPSAMPLE_SCALED, "new scaled smaller than previous; cannot upsample"); }
inline Error err_non_empty(const std::string& what) {
    return Error(E_NON_EMPTY_MINHASH, "Can only set \"" + what + "\" if the MinHash is empty");
}
inline Error err_invalid_dna(const std::string& kmer) {
    return Error(E_INVALID_DNA, "invalid DNA character in input k-mer: " + kmer);
}
inline Error err_internal(const std::string& msg) { return Error(E_INTERNAL, "internal error: \"" + msg + "\""); }

}  // namespace smg
