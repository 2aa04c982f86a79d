// Latency and issue cost of the wide traversal's one memory operation on gfx950: an 8-byte gather per lane from a
// cache-resident table (the 64-cell wide nodes: 512 B per node, a lane reads one 8-byte cell of a node).
// Per CU exactly w workgroups of 256 lanes (w waves per SIMD; see valu_rate.hip). Each lane chases a pointer chain
// through the table: idx = f(loaded word). Tables: 16 KiB (L1-resident), 1.25 MiB (dragon.vox's wide tree: L2),
// lanes of a wave spread over `spread` consecutive nodes (1 = one node, 512 B; 8 = 4 KiB: a tile's rays near a surface).
// Reported: ticks per dependent load as the wave sees it (latency incl. address arithmetic: 3 VALU) and per SIMD (/ w).
// Forms: global_load_dwordx2 with a 64-bit address (v_lshl_add_u64) and buffer_load_dwordx2 with a 32-bit offset.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip        Run: ./gather_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int kSteps = 2048;

typedef int v4i __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ __launch_bounds__(256) void chase(const uint2 *table, uint32_t n_nodes, uint32_t spread, uint64_t *stamps, uint32_t *out,
                                             int lds_words) {
    extern __shared__ float lds[];
    if (lds_words < 0) lds[threadIdx.x] = 1.0f;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t node = (blockIdx.x * 7u + (threadIdx.x >> 6) * 13u + lane % spread) % n_nodes;
    uint32_t cell = lane & 63u;
    v4i rsrc;
    {
        const uint64_t base = (uint64_t)table;
        rsrc.x = (int)(uint32_t)base;
        rsrc.y = (int)(uint32_t)(base >> 32);      // stride 0
        rsrc.z = (int)(n_nodes * 512u);            // bytes
        rsrc.w = 0x00020000;                       // raw buffer, dword data format off (gfx9: DST_SEL defaults), no swizzle
    }
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < kSteps; ++i) {
        uint2 v;
        const uint32_t idx = (node << 6) | cell;
        if constexpr (FORM == 0) {
            v = table[idx];
        } else {
            const uint32_t off = idx << 3;
            asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(off), "s"(rsrc) : "memory");
        }
        // the next node stays within the wave's window of `spread` nodes; the cell moves on: a dependent address
        node = v.x;
        cell = (cell + v.y) & 63u;
        acc += v.y;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = acc + node;
    if (lane == 0) stamps[gid >> 6] = t1 - t0;
}

template <int FORM>
static double run(int cus, int w, const uint2 *d_table, uint32_t n_nodes, uint32_t spread, uint64_t *d_st, uint32_t *d_out) {
    const int blocks = cus * w;
    size_t lds = (size_t)(160 * 1024) / (size_t)w;
    lds -= lds % 1024;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&chase<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL(chase<FORM>, dim3(blocks), dim3(256), lds, 0, d_table, n_nodes, spread, d_st, d_out, 1);
    (void)hipDeviceSynchronize();
    std::vector<uint64_t> st((size_t)blocks * 4);
    (void)hipMemcpy(st.data(), d_st, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::sort(st.begin(), st.end());
    return (double)st[st.size() / 2] / kSteps;
}

int main() {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = p.multiProcessorCount;
    uint64_t *d_st;
    uint32_t *d_out;
    (void)hipMalloc(&d_st, sizeof(uint64_t) * (size_t)cus * 8 * 4);
    (void)hipMalloc(&d_out, sizeof(uint32_t) * (size_t)cus * 8 * 256);
    printf("%s: ticks per dependent 8-byte gather, wave view / per SIMD; w = waves per SIMD\n", p.gcnArchName);
    const uint32_t sizes[2] = {32, 2435};  // nodes of 512 B: 16 KiB, 1.25 MiB
    for (int si = 0; si < 2; ++si) {
        const uint32_t n_nodes = sizes[si];
        for (uint32_t spread : {1u, 8u, 64u}) {
            if (spread > n_nodes) continue;
            // table: cell (node, c) -> x = another node of the same window of `spread` nodes, y = a cell increment
            std::vector<uint2> h((size_t)n_nodes * 64);
            uint32_t s = 12345u;
            for (uint32_t n = 0; n < n_nodes; ++n)
                for (uint32_t c = 0; c < 64; ++c) {
                    s = s * 1664525u + 1013904223u;
                    const uint32_t win = n / spread * spread;
                    const uint32_t span = std::min(spread, n_nodes - win);
                    h[(size_t)n * 64 + c] = make_uint2(win + (s >> 8) % span, 1u + ((s >> 20) & 15u));
                }
            uint2 *d_table;
            (void)hipMalloc(&d_table, h.size() * sizeof(uint2));
            (void)hipMemcpy(d_table, h.data(), h.size() * sizeof(uint2), hipMemcpyHostToDevice);
            for (int form = 0; form < 2; ++form) {
                printf("%-22s table %7.1f KiB spread %2u nodes:", form == 0 ? "global_load_dwordx2" : "buffer_load_dwordx2", n_nodes * 0.5, spread);
                for (int w : {1, 2, 4, 6, 8}) {
                    const double t = form == 0 ? run<0>(cus, w, d_table, n_nodes, spread, d_st, d_out)
                                               : run<1>(cus, w, d_table, n_nodes, spread, d_st, d_out);
                    printf("  w%d %7.1f/%6.1f", w, t, t / w);
                }
                printf("\n");
                fflush(stdout);
            }
            (void)hipFree(d_table);
        }
    }
    return 0;
}
