// sketch_long.hip -- the register-window sketch kernel for k = 65 .. 128 (see sketch.hip / sketch_kernel.hpp).
// Round 3 sent every k > 64 to the byte-wise kernel, 15 x slower at the same work (missing 4:
// src/core/src/signature.rs:246-306 has no k cliff).  A window of P + k - 1 = 143 bytes is 36 dwords per lane; the validity
// mask is 192 bits (kmer_core.hpp).  Compiled four times (-DSK_LONG_PART=0..3, 16 ksizes each: the Makefile) so that the 64
// fully unrolled instantiations build side by side instead of for six minutes in one unit.
#include "sketch_kernel.hpp"

#ifndef SK_LONG_PART
#error "compile with -DSK_LONG_PART=0..3"
#endif

namespace smg {

#define SK_CAT2(a, b) a##b
#define SK_CAT(a, b) SK_CAT2(a, b)
// ksizes 65 + 16 * part .. 80 + 16 * part
sketch_launch_fn SK_CAT(sparse_launcher_long_, SK_LONG_PART)(uint32_t k) {
    return sparse_launcher_from<64 + 16 * SK_LONG_PART>(k, std::make_integer_sequence<int, 16>());
}

}  // namespace smg
