// Issue-rate probe for gfx950: what ONE SIMD sustains per vector-instruction kind with 1, 2, 3, 4, 6 and 8 waves resident on it.
//
// Round 2's version claimed w waves per SIMD and did not have them (its in-kernel column read 0.81 ticks per wave64 instruction on a
// SIMD-32, which cannot be): its kernel held every operand array of every kind (96+ VGPRs: five waves at most) and nothing checked
// where the waves ran. This one makes residency a measured quantity:
//   * geometry: ONE workgroup of 4 * w waves per CU for w <= 4 (the workgroup claims 96 KiB of the CU's 160 KiB of LDS, so a second
//     one cannot join it), TWO workgroups of 4 * w / 2 waves per CU for w = 6, 8 (72 KiB each: two fit, three do not); the launch
//     has exactly as many workgroups as the chip holds, so every CU is full for the whole measurement;
//   * a kind's kernel holds only the operand arrays that kind uses (16 to 48 VGPRs: eight waves fit);
//   * every wave stores its HW_ID (s_getreg: SIMD, CU, shader array, shader engine), its XCC id and s_memtime at the start and the
//     end of its stream; the host groups the waves by physical SIMD, keeps the SIMDs that held exactly w waves whose streams all
//     overlapped, and reports, per kind and w,
//         simd = (latest end - earliest start of the SIMD's waves) / (w * instructions per wave): SIMD ticks per wave64 instruction,
//         wave = one wave's own ticks per instruction (its issue latency + arbitration),
//         ok   = how many of the chip's 1,024 SIMDs qualified (the rest held a different number of waves: reported, not used);
//     the launch's wall time is printed beside them (it contains the ramp and drain of the launch).
// Ticks are shader cycles (s_memtime); the clock under load is read from s_memrealtime (100 MHz). Nothing else runs meanwhile.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip        Run: ./valu_rate [out.json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

constexpr int kIters = 256, kUnroll = 16;

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum Kind {
    K_ADD_F32, K_MUL_F32, K_FMA_F32, K_PK_MUL_F32, K_PK_ADD_F32, K_PK_FMA_F32, K_CNDMASK_VCC, K_CMP_VCC, K_CMP_SGPR,
    K_CMP_CNDMASK, K_AND_B32, K_LSHLREV_B32, K_BFE_U32, K_ADD_U32, K_LSHL_ADD_U32, K_AND_OR_B32, K_CVT_FLR, K_CVT_F32_I32,
    K_CVT_UBYTE, K_RCP_F32, K_SQRT_F32, K_RSQ_F32, K_DIV_SCALE, K_DIV_FMAS, K_DIV_FIXUP, K_MAD_U64_U32, K_LSHLREV_B64,
    K_ADDC, K_MOV_B32, K_READFIRSTLANE, K_MIN_F32, K_MED3_F32, K_FLOOR_F32, K_SALU_AND_B64, K_SALU_ADD_U32, K_SALU_CSELECT,
    K_VALU_SALU_MIX, K_DEP_ADD_F32, K_DEP_FMA_F32, K_DEP_CMP_CNDMASK, K_EXEC_TOGGLE,
    K_MIN3_F32, K_FRACT_F32, K_CVT_U32_F32, K_MAD_U32_U24, K_MUL_U32_U24, K_MUL_LO_U32, K_BFI_B32, K_PERM_B32, K_OR3_B32,
    K_LSHL_OR_B32, K_XOR_B32, K_SUB_U32, K_MAX_U32, K_AND_LIT, K_CMPS_CNDMASK_E64, K_CMP_CNDMASK_E32, K_CNDMASK_E64, K_BR_EXECZ,
    K_BR_TAKEN, K_SAVEEXEC, K_LDEXP_F32, K_LSHRREV_V, K_ADD3_U32, K_ADD_LSHL_U32, K_SUB_F32, K_MAX_F32, K_MIN_U32, K_MUL_ADD_PAIR,
    K_FMA_MIX4, K_SALU_VALU_1_3, K_FULL_HALF, K_FULL_TRANS_3_1, K_HALF_HALF, K_INT_FLOAT, K_FULL_HALF_3_1, K_CVT_I32_F32, K_RNDNE_F32, K_MUL_HI_U32, K_COUNT
};
static const char *kNames[K_COUNT] = {
    "v_add_f32", "v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_cndmask_b32 (vcc)",
    "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 -> sgpr pair", "v_cmp + v_cndmask (2 insts)", "v_and_b32", "v_lshlrev_b32",
    "v_bfe_u32", "v_add_u32", "v_lshl_add_u32", "v_and_or_b32", "v_cvt_flr_i32_f32", "v_cvt_f32_i32", "v_cvt_f32_ubyte0",
    "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_mad_u64_u32",
    "v_lshlrev_b64", "v_add_co_u32 + v_addc_co_u32 (2 insts)", "v_mov_b32", "v_readfirstlane_b32", "v_min_f32", "v_med3_f32",
    "v_floor_f32", "s_and_b64", "s_add_u32", "s_cselect_b32", "v_add_f32 + s_add_u32 (2 insts)", "v_add_f32 dependent chain",
    "v_fma_f32 dependent chain", "v_cmp + v_cndmask dependent chain (2 insts)", "s_and_saveexec_b64 + s_or_b64 exec (2 insts)",
    "v_min3_f32", "v_fract_f32", "v_cvt_u32_f32", "v_mad_u32_u24", "v_mul_u32_u24", "v_mul_lo_u32", "v_bfi_b32", "v_perm_b32",
    "v_or3_b32", "v_lshl_or_b32", "v_xor_b32", "v_sub_u32", "v_max_u32", "v_and_b32 with 32-bit literal",
    "v_cmp_lt_u32 -> sgpr + v_cndmask_b32 e64 (2 insts)", "v_cmp_lt_u32 -> vcc + v_cndmask_b32 e32 (2 insts)",
    "v_cndmask_b32 e64 (fixed sgpr mask)", "s_cbranch_execz not taken", "s_cmp + s_cbranch_scc1 taken (2 insts)",
    "s_and_saveexec_b64 + s_mov exec (2 insts)", "v_ldexp_f32", "v_lshrrev_b32 by vgpr", "v_add3_u32", "v_add_lshl_u32",
    "v_sub_f32", "v_max_f32", "v_min_u32", "v_mul_f32 + v_add_f32 (2 insts)",
    "v_mul + v_floor + v_add + v_mul f32 (4 insts)", "s_and_b64 + 3 x v_add_f32 (4 insts)",
    "v_add_f32 + v_cndmask_b32 e64 (2 insts)", "3 x v_add_f32 + v_rcp_f32 (4 insts)", "v_floor_f32 + v_cndmask_b32 e64 (2 insts)",
    "v_and_b32 + v_add_f32 (2 insts)", "3 x v_add_f32 + v_cndmask_b32 e64 (4 insts)", "v_cvt_i32_f32", "v_rndne_f32", "v_mul_hi_u32"};
static const int kInstsPer[K_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 2, 2,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 2, 4, 4, 2, 4, 2, 2, 4, 1, 1, 1};


typedef float f2 __attribute__((ext_vector_type(2)));

// which operand arrays a kind's stream touches: 1 = a[] (f32), 2 = p[] (packed f32), 4 = u[] (u32), 8 = q[] (u64), 16 = b[] (second f32)
constexpr int regs_of(int k) {
    switch (k) {
        case K_PK_MUL_F32: case K_PK_ADD_F32: case K_PK_FMA_F32: return 2;
        case K_AND_B32: case K_LSHLREV_B32: case K_BFE_U32: case K_ADD_U32: case K_LSHL_ADD_U32: case K_AND_OR_B32: case K_CVT_FLR:
        case K_ADDC: case K_MOV_B32: case K_READFIRSTLANE: case K_CVT_U32_F32: case K_MAD_U32_U24: case K_MUL_U32_U24: case K_MUL_LO_U32:
        case K_BFI_B32: case K_PERM_B32: case K_OR3_B32: case K_LSHL_OR_B32: case K_XOR_B32: case K_SUB_U32: case K_MAX_U32: case K_AND_LIT:
        case K_CMPS_CNDMASK_E64: case K_CMP_CNDMASK_E32: case K_CNDMASK_E64: case K_LSHRREV_V: case K_ADD3_U32: case K_ADD_LSHL_U32:
        case K_MIN_U32: case K_CVT_I32_F32: case K_MUL_HI_U32: return 4;
        case K_MAD_U64_U32: case K_LSHLREV_B64: return 8;
        case K_SALU_AND_B64: case K_SALU_ADD_U32: case K_SALU_CSELECT: case K_EXEC_TOGGLE: case K_BR_EXECZ: case K_BR_TAKEN: case K_SAVEEXEC: return 0;
        case K_FULL_HALF: case K_HALF_HALF: case K_INT_FLOAT: case K_FULL_HALF_3_1: return 1 | 4;
        case K_FULL_TRANS_3_1: return 1 | 16;
        default: return 1;
    }
}

struct Stamp { uint64_t t0, t1, r0, r1; uint32_t hw_id, xcc_id; };

template <int KIND>
__global__ __launch_bounds__(1024) void probe(float *out, Stamp *stamps, float seed, int lds_words, uint32_t iseed) {
    extern __shared__ float lds[];
    if (lds_words < 0) lds[threadIdx.x] = seed;  // never: keeps the dynamic LDS claim alive
    constexpr int R = regs_of(KIND);
    float a[kUnroll], b[kUnroll];
    f2 p[kUnroll];
    uint32_t u[kUnroll];
    unsigned long long q[kUnroll];
    const float c = seed * 0.5f + (float)threadIdx.x * 1e-3f, d = seed * 0.25f + 1.0f;
    const uint32_t ci = iseed * 7u + threadIdx.x;
    const f2 c2 = {c, d};
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
        a[i] = (float)i + seed;
        b[i] = (float)i * 3.0f + seed;
        p[i] = f2{(float)i, (float)i + seed};
        u[i] = (uint32_t)i * 77u + ci;
        q[i] = (unsigned long long)i * 1234567ull + ci;
    }
    uint32_t s0 = iseed, s1 = iseed >> 1;
    unsigned long long sm = 0x5555555555555555ull;
    asm volatile("s_mov_b64 vcc, exec" : : : "vcc");
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    __syncthreads();   // the workgroup's waves start their streams together
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
#define ONE(i)                                                                                                          \
    if constexpr (KIND == K_ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));                         \
    else if constexpr (KIND == K_MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(d));                    \
    else if constexpr (KIND == K_FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(c), "v"(d));        \
    else if constexpr (KIND == K_PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));             \
    else if constexpr (KIND == K_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(c2));             \
    else if constexpr (KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(c2));         \
    else if constexpr (KIND == K_CNDMASK_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );    \
    else if constexpr (KIND == K_CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c) : "vcc");        \
    else if constexpr (KIND == K_CMP_SGPR) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(sm) : "v"(a[i]), "v"(c));       \
    else if constexpr (KIND == K_CMP_CNDMASK)                                                                           \
        asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(c), "v"(d) : "vcc");  \
    else if constexpr (KIND == K_AND_B32) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_LSHLREV_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));                      \
    else if constexpr (KIND == K_BFE_U32) asm volatile("v_bfe_u32 %0, %0, %1, 2" : "+v"(u[i]) : "v"(ci));                \
    else if constexpr (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_LSHL_ADD_U32) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(ci));      \
    else if constexpr (KIND == K_AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));         \
    else if constexpr (KIND == K_CVT_FLR) asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(u[i]) : "v"(c));                \
    else if constexpr (KIND == K_CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(ci));               \
    else if constexpr (KIND == K_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[i]) : "v"(ci));              \
    else if constexpr (KIND == K_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                 \
    else if constexpr (KIND == K_SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                               \
    else if constexpr (KIND == K_RSQ_F32) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));                                 \
    else if constexpr (KIND == K_DIV_SCALE) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %1" : "=v"(a[i]) : "v"(c), "v"(d) : "vcc"); \
    else if constexpr (KIND == K_DIV_FMAS) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d) : "vcc"); \
    else if constexpr (KIND == K_DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d)); \
    else if constexpr (KIND == K_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q[i]) : "v"(ci) : "vcc"); \
    else if constexpr (KIND == K_LSHLREV_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q[i]));                      \
    else if constexpr (KIND == K_ADDC)                                                                                  \
        asm volatile("v_add_co_u32 %0, vcc, %2, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(u[i]), "+v"(u[(i + 8) & 15]) : "v"(ci) : "vcc"); \
    else if constexpr (KIND == K_MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(ci));                       \
    else if constexpr (KIND == K_READFIRSTLANE) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(u[i]));       \
    else if constexpr (KIND == K_MIN_F32) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));                    \
    else if constexpr (KIND == K_MED3_F32) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));      \
    else if constexpr (KIND == K_FLOOR_F32) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));                             \
    else if constexpr (KIND == K_SALU_AND_B64) asm volatile("s_and_b64 %0, %0, exec" : "+s"(sm) : : "scc");              \
    else if constexpr (KIND == K_SALU_ADD_U32) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");        \
    else if constexpr (KIND == K_SALU_CSELECT) asm volatile("s_cselect_b32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");    \
    else if constexpr (KIND == K_VALU_SALU_MIX)                                                                         \
        asm volatile("v_add_f32 %0, %2, %0\n s_add_u32 %1, %1, %3" : "+v"(a[i]), "+s"(s0) : "v"(c), "s"(s1) : "scc");    \
    else if constexpr (KIND == K_DEP_ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[0]) : "v"(c));                \
    else if constexpr (KIND == K_DEP_FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(c), "v"(d));    \
    else if constexpr (KIND == K_DEP_CMP_CNDMASK)                                                                       \
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[0]) : "v"(c), "v"(d) : "vcc");  \
    else if constexpr (KIND == K_EXEC_TOGGLE)                                                                           \
        asm volatile("s_and_saveexec_b64 %0, vcc\n s_or_b64 exec, exec, %0" : "=s"(sm) : : "scc");                      \
    else if constexpr (KIND == K_MIN3_F32) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));      \
    else if constexpr (KIND == K_FRACT_F32) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));                             \
    else if constexpr (KIND == K_CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(c));                \
    else if constexpr (KIND == K_MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, 4, %1" : "+v"(u[i]) : "v"(ci));        \
    else if constexpr (KIND == K_MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, 5, %0" : "+v"(u[i]));                      \
    else if constexpr (KIND == K_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ci));             \
    else if constexpr (KIND == K_BFI_B32) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(ci));               \
    else if constexpr (KIND == K_PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));             \
    else if constexpr (KIND == K_OR3_B32) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));               \
    else if constexpr (KIND == K_LSHL_OR_B32) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(ci));        \
    else if constexpr (KIND == K_XOR_B32) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_MAX_U32) asm volatile("v_max_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_AND_LIT) asm volatile("v_and_b32 %0, 0x7f800000, %0" : "+v"(u[i]));                     \
    else if constexpr (KIND == K_CMPS_CNDMASK_E64)                                                                      \
        asm volatile("v_cmp_lt_u32 %1, %0, %2\n v_cndmask_b32 %0, %0, %2, %1" : "+v"(u[i]), "=&s"(sm) : "v"(ci));        \
    else if constexpr (KIND == K_CMP_CNDMASK_E32)                                                                       \
        asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ci) : "vcc");         \
    else if constexpr (KIND == K_CNDMASK_E64) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ci), "s"(sm)); \
    else if constexpr (KIND == K_BR_EXECZ) asm volatile("s_cbranch_execz 1f\n1:" : : : );                                \
    else if constexpr (KIND == K_BR_TAKEN) asm volatile("s_cmp_lg_u32 %0, 0x12345\n s_cbranch_scc1 1f\n s_nop 0\n1:" : : "s"(s1) : "scc"); \
    else if constexpr (KIND == K_SAVEEXEC) asm volatile("s_and_saveexec_b64 %0, exec\n s_mov_b64 exec, %0" : "=s"(sm) : : "scc"); \
    else if constexpr (KIND == K_LDEXP_F32) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(ci));               \
    else if constexpr (KIND == K_LSHRREV_V) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));             \
    else if constexpr (KIND == K_ADD3_U32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ci));             \
    else if constexpr (KIND == K_ADD_LSHL_U32) asm volatile("v_add_lshl_u32 %0, %0, %1, 2" : "+v"(u[i]) : "v"(ci));      \
    else if constexpr (KIND == K_SUB_F32) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));                    \
    else if constexpr (KIND == K_MAX_F32) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(c));                    \
    else if constexpr (KIND == K_MIN_U32) asm volatile("v_min_u32 %0, %1, %0" : "+v"(u[i]) : "v"(ci));                   \
    else if constexpr (KIND == K_MUL_ADD_PAIR) asm volatile("v_mul_f32 %0, %1, %0\n v_add_f32 %0, %2, %0" : "+v"(a[i]) : "v"(d), "v"(c)); \
    else if constexpr (KIND == K_FMA_MIX4)                                                                              \
        asm volatile("v_mul_f32 %0, %1, %0\n v_floor_f32 %0, %0\n v_add_f32 %0, %2, %0\n v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(d), "v"(c)); \
    else if constexpr (KIND == K_SALU_VALU_1_3)                                                                         \
        asm volatile("s_and_b64 %1, %1, exec\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0" : "+v"(a[i]), "+s"(sm) : "v"(c) : "scc");  \
    else if constexpr (KIND == K_FULL_HALF) asm volatile("v_add_f32 %0, %2, %0\n v_cndmask_b32 %1, %1, %3, %4" : "+v"(a[i]), "+v"(u[i]) : "v"(c), "v"(ci), "s"(sm)); \
    else if constexpr (KIND == K_FULL_TRANS_3_1) asm volatile("v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_rcp_f32 %1, %1" : "+v"(a[i]), "+v"(b[i]) : "v"(c)); \
    else if constexpr (KIND == K_HALF_HALF) asm volatile("v_floor_f32 %0, %0\n v_cndmask_b32 %1, %1, %2, %3" : "+v"(a[i]), "+v"(u[i]) : "v"(ci), "s"(sm)); \
    else if constexpr (KIND == K_INT_FLOAT) asm volatile("v_and_b32 %1, %3, %1\n v_add_f32 %0, %2, %0" : "+v"(a[i]), "+v"(u[i]) : "v"(c), "v"(ci)); \
    else if constexpr (KIND == K_FULL_HALF_3_1) asm volatile("v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_cndmask_b32 %1, %1, %3, %4" : "+v"(a[i]), "+v"(u[i]) : "v"(c), "v"(ci), "s"(sm)); \
    else if constexpr (KIND == K_CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(c)); \
    else if constexpr (KIND == K_RNDNE_F32) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])); \
    else if constexpr (KIND == K_MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ci));

        REP16(ONE)
#undef ONE
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)s0 + (float)(uint32_t)sm;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
        if constexpr (R & 1) s += a[i];
        if constexpr (R & 16) s += b[i];
        if constexpr (R & 2) s += p[i].x + p[i].y;
        if constexpr (R & 4) s += (float)u[i];
        if constexpr (R & 8) s += (float)(uint32_t)q[i];
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) stamps[gid >> 6] = Stamp{t0, t1, r0, r1, hw_id, xcc_id};
}

struct Result { double wave, simd, ghz, wall; int simds_ok, simds_seen; };

template <int KIND>
static Result run(int cus, int w, float *d_out, Stamp *d_st) {
    const int wgs_per_cu = w <= 4 ? 1 : 2;
    const int waves_per_wg = 4 * w / wgs_per_cu;
    const int blocks = cus * wgs_per_cu;
    const size_t lds = (wgs_per_cu == 1 ? 96 : 72) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves_per_wg), lds, 0, d_out, d_st, 1.0f, 1, 3u);   // warm: code object, clocks
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64 * waves_per_wg), lds, 0, d_out, d_st, 1.0f, 1, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    std::vector<Stamp> st((size_t)blocks * waves_per_wg);
    hipMemcpy(st.data(), d_st, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // HW_ID (gfx9 family): [3:0] wave slot, [5:4] SIMD, [7:6] pipe, [11:8] CU, [12] shader array, [15:13] shader engine
    std::map<uint32_t, std::vector<const Stamp *>> by_simd;
    for (const Stamp &s : st) {
        const uint32_t key = ((s.xcc_id & 0xfu) << 16) | (((s.hw_id >> 13) & 7u) << 12) | (((s.hw_id >> 12) & 1u) << 11) |
                             (((s.hw_id >> 8) & 0xfu) << 4) | ((s.hw_id >> 4) & 3u);
        by_simd[key].push_back(&s);
    }
    const double n = (double)kIters * kUnroll * kInstsPer[KIND];
    std::vector<double> simd, wave, ghz;
    for (const auto &kv : by_simd) {
        const auto &v = kv.second;
        if ((int)v.size() != w) continue;
        uint64_t first = ~0ull, last = 0, latest_start = 0, earliest_end = ~0ull;
        for (const Stamp *s : v) {
            first = std::min(first, s->t0); last = std::max(last, s->t1);
            latest_start = std::max(latest_start, s->t0); earliest_end = std::min(earliest_end, s->t1);
        }
        if (latest_start >= earliest_end) continue;   // the waves did not all run side by side
        simd.push_back((double)(last - first) / (n * w));
        for (const Stamp *s : v) {
            wave.push_back((double)(s->t1 - s->t0) / n);
            if (s->r1 > s->r0) ghz.push_back((double)(s->t1 - s->t0) / (double)(s->r1 - s->r0) * 0.1);
        }
    }
    const auto med = [](std::vector<double> &x) { if (x.empty()) return 0.0; std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
    const double g = med(ghz);
    return Result{med(wave), med(simd), g, (double)ms * 1e-3 * g * 1e9 / (n * w), (int)simd.size(), (int)by_simd.size()};
}

template <int KIND>
static void sweep(int cus, float *d_out, Stamp *d_st, std::string &json) {
    const int ws[6] = {1, 2, 3, 4, 6, 8};
    printf("%-46s", kNames[KIND]);
    json += std::string(json.size() > 1 ? ",\n" : "\n") + "  \"" + kNames[KIND] + "\": {";
    for (int k = 0; k < 6; ++k) {
        const Result r = run<KIND>(cus, ws[k], d_out, d_st);
        printf("  w%d %5.2f %5.2f (%4d) %5.2f", ws[k], r.simd, r.wave, r.simds_ok, r.wall);
        char buf[192];
        snprintf(buf, sizeof buf, "%s\"w%d\": {\"simd\": %.3f, \"wave\": %.3f, \"simds_ok\": %d, \"simds_seen\": %d, \"ghz\": %.3f, \"wall\": %.3f}", k ? ", " : "",
                 ws[k], r.simd, r.wave, r.simds_ok, r.simds_seen, r.ghz, r.wall);
        json += buf;
    }
    json += "}";
    printf("\n");
    fflush(stdout);
}

template <int K>
static void all(int cus, float *d_out, Stamp *d_st, std::string &json) {
    if constexpr (K < K_COUNT) {
        sweep<K>(cus, d_out, d_st, json);
        all<K + 1>(cus, d_out, d_st, json);
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, nominal %.2f GHz. Per w (waves resident per SIMD): SIMD ticks per instruction | one wave's ticks per instruction | "
           "(SIMDs that held exactly w overlapping waves, of %d) | by the launch's wall time\n", p.gcnArchName, cus, p.clockRate / 1e6, cus * 4);
    float *d_out;
    Stamp *d_st;
    hipMalloc(&d_out, sizeof(float) * (size_t)cus * 2 * 1024);
    hipMalloc(&d_st, sizeof(Stamp) * (size_t)cus * 2 * 16);
    std::string json = "{";
    all<0>(cus, d_out, d_st, json);
    json += "\n}\n";
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) { fputs(json.c_str(), f); fclose(f); }
    }
    hipFree(d_out);
    hipFree(d_st);
    return 0;
}
