// Issue-rate probe for gfx950 vector ALU instructions: one wave per SIMD on every CU runs a long unrolled loop of
// one instruction kind over independent accumulators; prints cycles per wave-instruction derived from the wall
// time at the clock reported by the device. Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096, kUnroll = 16;

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, float seed) {
    f2 acc[kUnroll];
    float m = seed + threadIdx.x;
    f2 mm = {m, m};
    f2 c = {seed * 0.5f, seed * 0.25f};
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) acc[i] = f2{(float)i, (float)i + seed};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (KIND == 0) {  // scalar fma on .x only
                acc[i].x = __builtin_fmaf(m, c.x, acc[i].x);
            } else if (KIND == 1) {  // packed fma, plain operands
                acc[i] = __builtin_elementwise_fma(mm, c, acc[i]);
            } else if (KIND == 2) {  // packed fma with a broadcast first operand (op_sel_hi:[0,1,1])
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(mm), "v"(c));
            } else if (KIND == 3) {  // packed add
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(c));
            } else if (KIND == 4) {  // compare + select pair
                asm volatile("v_cmp_eq_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, 1.0, vcc" : "=v"(acc[i].x) : "v"(m), "v"(c.x) : "vcc");
            } else if (KIND == 5) {  // plain add
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(acc[i].x) : "v"(c.x));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, float *d_out, int cus, double ghz, int instr_per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<KIND>, dim3(cus), dim3(256), 0, 0, d_out, 1.0f);
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(probe<KIND>, dim3(cus), dim3(256), 0, 0, d_out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_launch_s = ms / 1e3 / reps;
    const double n = (double)kIters * kUnroll * instr_per_iter;  // wave-instructions per wave (one wave per SIMD)
    printf("%-34s %8.3f us/launch  %6.2f cycles per wave-instruction at %.2f GHz\n", name, per_launch_s * 1e6,
           per_launch_s * ghz * 1e9 / n, ghz);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("%s: %d CUs, %.2f GHz\n", p.gcnArchName, cus, ghz);
    float *d_out;
    hipMalloc(&d_out, sizeof(float) * cus * 256);
    run<0>("v_fma_f32", d_out, cus, ghz, 1);
    run<1>("v_pk_fma_f32", d_out, cus, ghz, 1);
    run<2>("v_pk_fma_f32 op_sel_hi:[0,1,1]", d_out, cus, ghz, 1);
    run<3>("v_pk_add_f32", d_out, cus, ghz, 1);
    run<4>("v_cmp_eq_u32 + v_cndmask_b32", d_out, cus, ghz, 2);
    run<5>("v_add_f32", d_out, cus, ghz, 1);
    hipFree(d_out);
    return 0;
}
