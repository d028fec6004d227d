// Lane sets of a wave as 64-bit masks in scalar registers (gfx950, wave64).
//
// A per-lane flag that is tested, merged and carried around a loop costs VALU instructions every time (select 0/1, compare
// with 0, and/or); the same set as a wave mask costs scalar ones, which issue beside the vector work.  `mask_of` of a PLAIN
// compare is the compare's own result register; of a compound condition the compiler materialises the flag first -- so take
// the masks of the compares and combine them with & | ~.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smg {

__device__ __forceinline__ uint64_t mask_of(bool lane_pred) { return __builtin_amdgcn_ballot_w64(lane_pred); }
// the mask as a lane predicate again: `if (lanes_of(m))` is one s_and_saveexec
__device__ __forceinline__ bool lanes_of(uint64_t wave_mask) { return __builtin_amdgcn_inverse_ballot_w64(wave_mask); }
// the value as the registers hold it, whatever wrote it (lanes a masked load skipped keep what they had; their compare
// bits must be masked off by the caller)
__device__ __forceinline__ void opaque(unsigned long long& v) { asm volatile("" : "+v"(v)); }

// a value every lane of the wave holds alike, moved to scalar registers (address arithmetic and compares on it turn scalar)
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v);
}

}  // namespace smg
