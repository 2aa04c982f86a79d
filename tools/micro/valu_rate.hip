// Issue-rate probe for gfx950: what one SIMD sustains per instruction kind at 1, 2, 4, 6 and 8 resident waves.
//
// Every CU gets exactly `w` workgroups of 256 lanes (4 waves = one per SIMD): the launch has CUs * w workgroups and each
// claims floor(160 KiB / w) of LDS, so no CU can hold more than w and every CU must hold w. A wave runs a long unrolled
// loop of ONE instruction kind over 16 independent registers (or one register: the dependent-chain rows) and stamps
// s_memtime around it. Reported per kind and w:
//     wave  = ticks one wave needs per instruction (its own view: latency + arbitration)
//     simd  = wave / w = SIMD time per instruction = 1 / throughput -- the figure an issue roofline is built from
// (ticks = shader cycles; the in-kernel clock is printed from s_memrealtime, 100 MHz.) Nothing else runs meanwhile.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip        Run: ./valu_rate [out.json]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

constexpr int kIters = 512, kUnroll = 16;

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
    K_FMA_MIX4, K_SALU_VALU_1_3, K_COUNT
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
    "v_mul + v_floor + v_add + v_mul f32 (4 insts)", "s_and_b64 + 3 x v_add_f32 (4 insts)"};
static const int kInstsPer[K_COUNT] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 2, 2,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 2, 4, 4};

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, uint64_t *stamps, float seed, int lds_words, uint32_t iseed) {
    extern __shared__ float lds[];
    if (lds_words < 0) lds[threadIdx.x] = seed;  // never: keeps the dynamic LDS claim alive
    float a[kUnroll];
    f2 p[kUnroll];
    uint32_t u[kUnroll];
    unsigned long long q[kUnroll];
    const float c = seed * 0.5f + (float)threadIdx.x * 1e-3f, d = seed * 0.25f + 1.0f;
    const uint32_t ci = iseed * 7u + threadIdx.x;
    const f2 c2 = {c, d};
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
        a[i] = (float)i + seed;
        p[i] = f2{(float)i, (float)i + seed};
        u[i] = (uint32_t)i * 77u + ci;
        q[i] = (unsigned long long)i * 1234567ull + ci;
    }
    uint32_t s0 = iseed, s1 = iseed >> 1;
    unsigned long long sm = ~0ull;
    asm volatile("s_mov_b64 vcc, exec" : : : "vcc");
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
        asm volatile("s_and_b64 %1, %1, exec\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %2, %0" : "+v"(a[i]), "+s"(sm) : "v"(c) : "scc");
        REP16(ONE)
#undef ONE
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)s0 + (float)(uint32_t)sm;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i] + (float)(uint32_t)q[i];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (gid >> 6)] = t1 - t0;
        stamps[2 * (gid >> 6) + 1] = r1 - r0;
    }
}

struct Result { double wave, simd, ghz, wall; };  // wall: launch time (hipEvents) x in-kernel clock / (instructions per wave x w): SIMD ticks per instruction by the wall clock

template <int KIND>
static Result run(int cus, int w, float *d_out, uint64_t *d_st) {
    const int blocks = cus * w;
    size_t lds = (size_t)(160 * 1024) / (size_t)w;
    lds -= lds % 1024;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), lds, 0, d_out, d_st, 1.0f, 1, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), lds, 0, d_out, d_st, 1.0f, 1, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    std::vector<uint64_t> st((size_t)blocks * 4 * 2);
    hipMemcpy(st.data(), d_st, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::vector<double> ticks, ghz;
    for (size_t i = 0; i < st.size() / 2; ++i) {
        ticks.push_back((double)st[2 * i]);
        if (st[2 * i + 1]) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);
    }
    std::sort(ticks.begin(), ticks.end());
    std::sort(ghz.begin(), ghz.end());
    const double n = (double)kIters * kUnroll * kInstsPer[KIND];
    const double wave = ticks[ticks.size() / 2] / n;
    const double g = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
    return Result{wave, wave / w, g, (double)ms * 1e-3 * g * 1e9 / (n * w)};
}

template <int KIND>
static void sweep(int cus, float *d_out, uint64_t *d_st, std::string &json) {
    const int ws[5] = {1, 2, 4, 6, 8};
    printf("%-46s", kNames[KIND]);
    json += std::string(json.size() > 1 ? ",\n" : "\n") + "  \"" + kNames[KIND] + "\": {";
    for (int k = 0; k < 5; ++k) {
        const Result r = run<KIND>(cus, ws[k], d_out, d_st);
        printf("  w%d %6.2f/%5.2f/%5.2f", ws[k], r.wave, r.simd, r.wall);
        char buf[128];
        snprintf(buf, sizeof buf, "%s\"w%d\": {\"wave\": %.3f, \"simd\": %.3f, \"ghz\": %.3f, \"wall\": %.3f}", k ? ", " : "", ws[k], r.wave, r.simd, r.ghz, r.wall);
        json += buf;
    }
    json += "}";
    printf("\n");
    fflush(stdout);
}

template <int K>
static void all(int cus, float *d_out, uint64_t *d_st, std::string &json) {
    if constexpr (K < K_COUNT) {
        sweep<K>(cus, d_out, d_st, json);
        all<K + 1>(cus, d_out, d_st, json);
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, nominal %.2f GHz. Columns: ticks per instruction as one wave sees it / per SIMD (= wave / w), "
           "w = waves resident per SIMD\n", p.gcnArchName, cus, p.clockRate / 1e6);
    float *d_out;
    uint64_t *d_st;
    hipMalloc(&d_out, sizeof(float) * (size_t)cus * 8 * 256);
    hipMalloc(&d_st, sizeof(uint64_t) * (size_t)cus * 8 * 4 * 2);
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
