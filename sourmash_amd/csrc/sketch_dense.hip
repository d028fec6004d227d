// sketch_dense.hip -- the register-window kernel in its per-position form (one hash per k-mer start: kmerminhash_seq_to_hashes,
// src/core/src/ffi/minhash.rs:63-99 over src/core/src/signature.rs:246-306) for EVERY k = 1 .. SK_FAST_MAX_K = 88 (longer k-mers: sketch_words.hip).  Round 4
// instantiated that form for k = 21, 31, 51 only; every other ksize took the byte-wise kernel, 15-30 x slower at the same work
// (Missing 3).  Compiled six times (-DSK_DENSE_PART=0..5, up to 16 ksizes each: the Makefile) so that the fully
// unrolled instantiations build side by side.
#include "sketch_kernel.hpp"

#ifndef SK_DENSE_PART
#error "compile with -DSK_DENSE_PART=0..5"
#endif

namespace smg {

#define SK_CAT2(a, b) a##b
#define SK_CAT(a, b) SK_CAT2(a, b)
// ksizes 1 + 16 * part .. min(16 + 16 * part, SK_FAST_MAX_K)
sketch_launch_fn SK_CAT(dense_launcher_, SK_DENSE_PART)(uint32_t k) {
    return dense_launcher_from<16 * SK_DENSE_PART>(k, std::make_integer_sequence<int, sk_part_size(16 * SK_DENSE_PART)>());
}

}  // namespace smg
