// sketch_long.hip -- the register-window sketch kernel for k = 65 .. SK_FAST_MAX_K = 88 (see sketch.hip / sketch_kernel.hpp).
// Round 3 sent every k > 64 to the byte-wise kernel, 15 x slower at the same work (missing 4:
// src/core/src/signature.rs:246-306 has no k cliff); round 4 instantiated this kernel up to k = 128; round 5 hands k >= 89 -- where
// a window of P + k - 1 bytes plus both strands' words no longer fit 256 registers at two waves per SIMD -- to the run-time-k
// kernel of sketch_words.hip.  Compiled twice (-DSK_LONG_PART=0..1, up to 16 ksizes each: the Makefile) so that the fully
// unrolled instantiations build side by side.
#include "sketch_kernel.hpp"

#ifndef SK_LONG_PART
#error "compile with -DSK_LONG_PART=0..1"
#endif

namespace smg {

#define SK_CAT2(a, b) a##b
#define SK_CAT(a, b) SK_CAT2(a, b)
// ksizes 65 + 16 * part .. min(80 + 16 * part, SK_FAST_MAX_K)
sketch_launch_fn SK_CAT(sparse_launcher_long_, SK_LONG_PART)(uint32_t k) {
    return sparse_launcher_from<64 + 16 * SK_LONG_PART>(k, std::make_integer_sequence<int, sk_part_size(64 + 16 * SK_LONG_PART)>());
}

}  // namespace smg
