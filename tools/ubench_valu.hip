// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the integer VALU
// instructions the sketch kernel is made of, on gfx950.  Build: hipcc --offload-arch=gfx950 -O3
// tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.  Results feed DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"

#define REP8(x) x x x x x x x x
#define ITERS 16384

// each kernel: 8 independent dependency chains per lane, ITERS * 8 * 8 instructions per wave
#define DEFK(NAME, ASMSTR)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                    \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,     \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                         \
        uint32_t b = seed | 1, c = seed * 2654435761u;                                             \
        for (int i = 0; i < ITERS; ++i) {                                                          \
            REP8(asm volatile(ASMSTR(0) ASMSTR(1) ASMSTR(2) ASMSTR(3) ASMSTR(4) ASMSTR(5)         \
                              ASMSTR(6) ASMSTR(7)                                                  \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),      \
                                "+v"(a6), "+v"(a7)                                                 \
                              : "v"(b), "v"(c) : "vcc", "s10", "s11");)                                                  \
        }                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
    }

#define S_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define S_MUL_HI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define S_MUL_U24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define S_MAD_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define S_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define S_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define S_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define S_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define S_XOR3(i) "v_xor3_b32 %" #i ", %" #i ", %8, %9\n"
#define S_CNDMASK(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define S_MAD_U32(i) "v_mad_u32_u16 %" #i ", %" #i ", %8, %9\n"

#define S_AND(i) "v_and_b32_e32 %" #i ", %" #i ", %8\n"
#define S_OR(i) "v_or_b32_e32 %" #i ", %" #i ", %8\n"
#define S_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 3, %" #i "\n"
#define S_LSHR(i) "v_lshrrev_b32_e32 %" #i ", 3, %" #i "\n"
#define S_SUB(i) "v_sub_u32_e32 %" #i ", %" #i ", %8\n"
#define S_MOV(i) "v_mov_b32_e32 %" #i ", %8\n"
#define S_ADDCO(i) "v_add_co_u32_e32 %" #i ", vcc, %" #i ", %8\n"
#define S_ADDC(i) "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define S_CND32(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define S_CND64S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define S_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define S_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define S_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define S_ADD64E(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n"
#define S_XOR64E(i) "v_xor_b32_e64 %" #i ", %" #i ", %8\n"
#define S_MULU24E32(i) "v_mul_u32_u24_e32 %" #i ", %" #i ", %8\n"
#define S_CMP32(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n"
#define S_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
DEFK(k_and_e32, S_AND) DEFK(k_or_e32, S_OR) DEFK(k_lshl_e32, S_LSHL) DEFK(k_lshr_e32, S_LSHR) DEFK(k_sub_e32, S_SUB)
DEFK(k_mov_e32, S_MOV) DEFK(k_addco_e32, S_ADDCO) DEFK(k_addc_e32, S_ADDC) DEFK(k_cnd_e32, S_CND32) DEFK(k_cnd_e64_sgpr, S_CND64S)
DEFK(k_and_or, S_ANDOR) DEFK(k_lshl_or, S_LSHLOR) DEFK(k_or3, S_OR3) DEFK(k_bfe, S_BFE) DEFK(k_add_e64, S_ADD64E) DEFK(k_xor_e64, S_XOR64E)
DEFK(k_mul_u24_e32, S_MULU24E32) DEFK(k_cmp_lt_u32_e32, S_CMP32) DEFK(k_bitop3_xor3, S_BITOP3)
DEFK(k_mul_lo, S_MUL_LO)
DEFK(k_mul_hi, S_MUL_HI)
DEFK(k_mul_u24, S_MUL_U24)
DEFK(k_mad_u24, S_MAD_U24)
DEFK(k_add, S_ADD)
DEFK(k_add3, S_ADD3)
DEFK(k_xor, S_XOR)
DEFK(k_perm, S_PERM)
DEFK(k_alignbit, S_ALIGNBIT)
DEFK(k_lshladd, S_LSHLADD)
DEFK(k_cndmask, S_CNDMASK)

// 64-bit forms: 4 chains of register pairs
#define DEFK64(NAME, ASMSTR)                                                                       \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {                    \
        uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,     \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                                         \
        uint64_t b = ((uint64_t)seed << 32) | 12345u;                                              \
        uint32_t c = seed | 1;                                                                     \
        for (int i = 0; i < ITERS; ++i) {                                                          \
            REP8(asm volatile(ASMSTR(0) ASMSTR(1) ASMSTR(2) ASMSTR(3) ASMSTR(4) ASMSTR(5)         \
                              ASMSTR(6) ASMSTR(7)                                                  \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),      \
                                "+v"(a6), "+v"(a7)                                                 \
                              : "v"(b), "v"(c) : "vcc", "s10", "s11");)                                                  \
        }                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7); \
    }
#define S_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %9, %9, %" #i "\n"
#define S_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %8\n"
#define S_LSHL64(i) "v_lshlrev_b64 %" #i ", 7, %" #i "\n"
#define S_LSHR64(i) "v_lshrrev_b64 %" #i ", 7, %" #i "\n"
#define S_CMP64(i) "v_cmp_lt_u64 vcc, %" #i ", %8\n"
#define S_MOV64(i) "v_mov_b64 %" #i ", %8\n"
DEFK64(k_mad_u64_u32, S_MAD64)
DEFK64(k_lshl_add_u64, S_LSHLADD64)
DEFK64(k_lshlrev_b64, S_LSHL64)
DEFK64(k_lshrrev_b64, S_LSHR64)
DEFK64(k_cmp_lt_u64, S_CMP64)

typedef void (*kern_t)(uint32_t*, uint32_t);

static void run(const char* name, kern_t k, int waves_per_simd) {
    const int blocks_per_cu = waves_per_simd;          // 256 threads = 4 waves = 1 wave per SIMD
    const int blocks = 256 * blocks_per_cu;
    uint32_t* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)ITERS * 8 * 8;
    const double clk = 2.4e9;   // nominal; report both time and cycles at nominal clock
    const double cyc_per_inst_per_simd = ms * 1e-3 * clk / (insts_per_wave * waves_per_simd);
    printf("%-18s waves/SIMD=%d  %8.3f ms  %6.2f cycles/wave-instr/SIMD (at 2.4 GHz nominal)\n", name, waves_per_simd, ms,
           cyc_per_inst_per_simd);
    hipFree(d);
}

int main() {
#define R(k) run(#k, k, 2); run(#k, k, 4);
    R(k_and_e32) R(k_or_e32) R(k_lshl_e32) R(k_lshr_e32) R(k_sub_e32) R(k_mov_e32) R(k_addco_e32) R(k_addc_e32) R(k_cnd_e32) R(k_cnd_e64_sgpr) R(k_and_or) R(k_lshl_or) R(k_or3) R(k_bfe) R(k_add_e64) R(k_xor_e64) R(k_mul_u24_e32) R(k_cmp_lt_u32_e32) R(k_bitop3_xor3) R(k_add) R(k_xor) R(k_add3) R(k_lshladd) R(k_perm) R(k_alignbit) R(k_cndmask)
    R(k_mul_u24) R(k_mad_u24) R(k_mul_lo) R(k_mul_hi) R(k_mad_u64_u32) R(k_lshl_add_u64)
    R(k_lshlrev_b64) R(k_lshrrev_b64) R(k_cmp_lt_u64)
    return 0;
}
