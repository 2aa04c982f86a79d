// vrt_test.hip -- TEST SUPPORT, not product: libvrt_hip_test.so. What the parity suite needs to look INSIDE the dispatch layer
// without the product library exporting hooks for it:
//   * device probes of the arithmetic the kernels rely on (correctly rounded / and sqrt, no contraction, the in-range 1/x, sqrt and
//     x/PI forms, the polynomial exp / sin / cos / pow conventions): the same inline functions the kernels call, compiled from the
//     same headers with the same flags;
//   * host-only views of the layouts the uploader builds (record array, wide cells), of the edit patches, of the per-projection
//     ray tables and of the dispatcher's "world is empty outside wide root 0" analysis: the same sources (vrt_layout.cpp,
//     vrt_raygen.cpp) linked a second time.
// libvrt_hip.so exports none of this (tests/test_abi.py checks that).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../../include/vrt.h"
#include "../vrt_common.hip.h"
#include "../vrt_full.hip.h"
#include "../vrt_layout.h"
#include "../vrt_sched.hip.h"

namespace vrt {
// exactness probe for the arithmetic contract: out[i] = op(x[i], y[i])
__global__ void math_probe_kernel(int op, const float *x, const float *y, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], r = 0.0f;
    switch (op) {
        case 0: r = a / b; break;
        case 1: r = __builtin_sqrtf(a); break;
        case 2: r = 1.0f / __builtin_sqrtf(a); break;
        case 3: r = __builtin_floorf(a); break;
        case 4: r = __builtin_rintf(a); break;
        case 5: r = a * b + 1.0f; break;          // must NOT be fused
        case 6: r = det_expf(a); break;
        case 7: r = (float)(int)a; break;
        case 8: r = a + b; break;
        case 9: r = a * b; break;
        case 30: r = rcp_inrange(a); break;
        case 31: r = sqrt_inrange(a); break;
        case 32: r = div_pi_inrange(a); break;
        default: break;
    }
    out[i] = r;
}

namespace full {
// exactness probe for the conventions above (ops 10..): out[i] = op(x[i], y[i])
__global__ void math_probe_full_kernel(int op, const float *x, const float *y, float *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[i], b = y[i], r = 0.0f, s, c;
    switch (op) {
        case 10: det_sincos(a, s, c); r = s; break;
        case 11: det_sincos(a, s, c); r = c; break;
        case 12: r = det_powf(a, b); break;
        case 13: { int q; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(q) : "v"(a)); r = (float)q; } break;
        case 14: r = (float)__float_as_uint(a) / 4294967296.0f; break;  // rand(): uint -> float, RNE
        case 15: r = unorm_of(a); break;                                // must equal a / 255.0f for the 256 byte values
        default: break;
    }
    out[i] = r;
}

}  // namespace full
}  // namespace vrt

namespace vrt_internal {
bool build_ray_table(const float *m, int W, int H, std::vector<float> &tab, float &z_out);
bool view_matrix_in_range(const float *m);
}
using vrt_internal::build_ray_table;
using vrt_internal::view_matrix_in_range;

#define VRT_HIP(c, call)                       \
    do {                                       \
        if ((call) != hipSuccess) return VRT_E_HIP; \
    } while (0)

extern "C" {

// Arithmetic-contract probe (math_probe_kernel / math_probe_full_kernel): out[i] = op(x[i], y[i]) on `device`; host arrays, synchronous.
int vrt_test_math(int device, int op, const float *x, const float *y, float *out, int n) {
    if (!x || !y || !out || n < 1) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(device));
    void *c = nullptr; (void)c;
    hipStream_t stream = nullptr;
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    struct Free { float *&a, *&b, *&o; ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(o); } } free_on_exit{dx, dy, dout};
    VRT_HIP(c, hipMalloc((void **)&dx, bytes));
    VRT_HIP(c, hipMalloc((void **)&dy, bytes));
    VRT_HIP(c, hipMalloc((void **)&dout, bytes));
    VRT_HIP(c, hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, stream));
    VRT_HIP(c, hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, stream));
    if (op >= 10 && op < 30)
        hipLaunchKernelGGL(vrt::full::math_probe_full_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, dx, dy, dout, n);
    else
        hipLaunchKernelGGL(vrt::math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, dx, dy, dout, n);
    VRT_HIP(c, hipGetLastError());
    VRT_HIP(c, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, stream));
    VRT_HIP(c, hipStreamSynchronize(stream));
    return VRT_OK;
}

// Host-only (tests): the ray-generation table the dispatcher would build for this inverse projection and frame shape.
// Returns 1 and fills out_x[width], out_y[height], out_z when the projection has a table, 0 when it has none.
int vrt_test_ray_table(const float inv_projection[16], int width, int height, float *out_x, float *out_y, float *out_z) {
    if (!inv_projection || width < 1 || height < 1 || !out_x || !out_y || !out_z) return VRT_E_INVALID;
    std::vector<float> tab;
    float z = 0.0f;
    if (!build_ray_table(inv_projection, width, height, tab, z)) return 0;
    std::memcpy(out_x, tab.data(), (size_t)width * sizeof(float));
    std::memcpy(out_y, tab.data() + width, (size_t)height * sizeof(float));
    *out_z = z;
    return 1;
}
// ... and whether an inverse view matrix keeps the second normalisation in range (view_matrix_in_range())
int vrt_test_view_in_range(const float inv_view[16]) { return inv_view ? (view_matrix_in_range(inv_view) ? 1 : 0) : VRT_E_INVALID; }

// Host-only (tests): what the dispatcher would tell the kernels about wide root 0 for this tree, these world bounds and
// this eye: out[0] = root0_only, out[1] = log2 of the chosen root's side, out[2..4] = its minimum corner, out[5] = log2 of
// the side of build_wide()'s root. Returns 0, VRT_E_MALFORMED, or VRT_E_STATE when the scene has no wide form.
int vrt_test_root0(const uint8_t *texels, size_t used_bytes, const int32_t wmin[3], const int32_t wmax[3], const int32_t eye[3],
                    int32_t out[6]) {
    if (!wmin || !wmax || !eye || !out) return VRT_E_INVALID;
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    vrt::WideTree wt;
    if (vrt::has_unit_internal_node(lay.records, wmin, wmax) || !vrt::build_wide(lay.records, wmin, wmax, wt, err) || wt.roots.empty())
        return VRT_E_STATE;
    uint32_t node = wt.roots[0].node;
    int shift = wt.roots[0].shift, mn[3] = {wt.roots[0].origin[0], wt.roots[0].origin[1], wt.roots[0].origin[2]};
    out[5] = shift;
    out[0] = vrt::content_only_in_root0(lay.records, wt) ? 1 : 0;
    const int eyes[1][3] = {{eye[0], eye[1], eye[2]}};
    if (out[0]) vrt::tighten_root0(wt, eyes, 1, vrt::v3::kAnchorShift, node, shift, mn);
    out[1] = shift; out[2] = mn[0]; out[3] = mn[1]; out[4] = mn[2];
    return VRT_OK;
}

// Host-only check of the wide layout (tests without a GPU): builds it for the texel stream and world
// bounds and answers n point queries through it. out: n * 8 words = w0, w1, mn[3], mx[3].
// stats (optional): wide nodes, roots. Returns 0, VRT_E_MALFORMED, or VRT_E_STATE when the scene has no wide form.
int vrt_test_wide_find(const uint8_t *texels, size_t used_bytes, const int32_t wmin[3], const int32_t wmax[3],
                        const int32_t *points, size_t n, uint32_t *out, uint32_t *stats) {
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    vrt::WideTree wt;
    if (!vrt::build_wide(lay.records, wmin, wmax, wt, err)) return VRT_E_STATE;
    if (stats) { stats[0] = wt.n_nodes; stats[1] = (uint32_t)wt.roots.size(); }
    for (size_t i = 0; i < n; ++i) {
        uint32_t w0, w1;
        int mn[3], mx[3];
        (void)vrt::wide_find_host(lay.records, wt, wmin, wmax, points + 3 * i, w0, w1, mn, mx);
        uint32_t *o = out + 8 * i;
        o[0] = w0; o[1] = w1;
        for (int k = 0; k < 3; ++k) { o[2 + k] = (uint32_t)mn[k]; o[5 + k] = (uint32_t)mx[k]; }
    }
    return VRT_OK;
}

// Host-only check of the edit patch (no device): lays out the tree before and after an edit of voxel (x, y, z) from
// their texel streams, patches the "before" structures with the sub-tree taken from the "after" ones, and counts the
// query points whose lookup (leaf words + node box), through the wide layout and through the records alone, differs
// between the patched and the freshly built structures. info: [0] depth of the node replaced (0: no patchable
// ancestor, nothing compared), [1] records appended, [2] wide cells appended, [3] 1 when the stream's texel count
// tracked by the patch equals the "after" stream's. sparse: the sub-tree carries only the nodes that contain the
// voxel, everything else as kKeep records. Returns the number of differing points or a negative code.
long vrt_test_patch_check(const uint8_t *before, size_t before_bytes, const uint8_t *after, size_t after_bytes,
                           const int32_t wmin[3], const int32_t wmax[3], int x, int y, int z, const int32_t *points, size_t n,
                           uint32_t *info, int sparse) {
    vrt::Layout lb, la;
    std::string err;
    if (!vrt::build_layout(before, before_bytes, lb, err) || !vrt::build_layout(after, after_bytes, la, err)) return VRT_E_MALFORMED;
    vrt::WideTree wb, wa;
    const bool wide_b = !vrt::has_unit_internal_node(lb.records, wmin, wmax) && vrt::build_wide(lb.records, wmin, wmax, wb, err);
    const bool wide_a = !vrt::has_unit_internal_node(la.records, wmin, wmax) && vrt::build_wide(la.records, wmin, wmax, wa, err);
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    const int p[3] = {x, y, z};
    vrt::PatchSite site;
    std::vector<vrt::Record> sub;
    int max_depth = 15;
    bool have = false;
    while (max_depth >= 1 && vrt::plan_patch(lb.records, wb, wide_b, wmin, wmax, p, max_depth, site)) {
        if (vrt::extract_subtree(la.records, site.path, site.depth, sub, sparse ? p : nullptr, wmin, wmax)) { have = true; break; }
        max_depth = site.depth - 1;
    }
    if (!have) return 0;
    const size_t n_rec = lb.records.size(), n_cells = wb.cells.size();
    vrt::PatchRanges rg;
    if (!vrt::apply_patch(lb.records, wb, wide_b, site, sub.data(), sub.size(), rg, err)) return VRT_E_MALFORMED;
    bool wide_p = wide_b;
    if (wide_b && rg.wide_invalid) wide_p = !vrt::has_unit_internal_node(lb.records, wmin, wmax) && vrt::build_wide(lb.records, wmin, wmax, wb, err);
    if (info) {
        info[0] = (uint32_t)site.depth;
        info[1] = (uint32_t)(lb.records.size() - n_rec);
        info[2] = (uint32_t)(wb.cells.size() > n_cells ? wb.cells.size() - n_cells : 0);
        info[3] = ((long)(before_bytes / 4) + rg.texel_delta == (long)(after_bytes / 4)) ? 1u : 0u;
    }
    if (wide_p != wide_a) return VRT_E_STATE;
    const vrt::WideTree none;
    long bad = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0 && !wide_p) continue;
            uint32_t a0, a1, b0, b1;
            int amn[3], amx[3], bmn[3], bmx[3];
            const int ka = vrt::wide_find_host(lb.records, pass == 0 ? wb : none, wmin, wmax, points + 3 * i, a0, a1, amn, amx);
            const int kb = vrt::wide_find_host(la.records, pass == 0 ? wa : none, wmin, wmax, points + 3 * i, b0, b1, bmn, bmx);
            bool same = ka == kb && a0 == b0 && a1 == b1;
            for (int k = 0; k < 3; ++k) same = same && amn[k] == bmn[k] && amx[k] == bmx[k];
            if (!same) { ++bad; break; }
        }
    }
    return bad;
}

// Host-only: what the dispatcher concludes about a tree before it runs VRT_MODE_FULL without a ray stack (vrt::tree_is_opaque):
// 1 / 0, or VRT_E_MALFORMED.
int vrt_test_tree_is_opaque(const uint8_t *texels, size_t used_bytes) {
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    return vrt::tree_is_opaque(lay.records) ? 1 : 0;
}

// Host-only view of the device layout for tests that run without a GPU:
// writes up to cap records (8 bytes each) and returns the record count, or <0.
long vrt_test_build_layout(const uint8_t *texels, size_t used_bytes, uint32_t *records_out, size_t cap_records,
                            vrt_scene_info *info) {
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    if (records_out)
        for (size_t i = 0; i < lay.records.size() && i < cap_records; ++i) {
            records_out[2 * i] = lay.records[i].w0;
            records_out[2 * i + 1] = lay.records[i].w1;
        }
    if (info) {
        std::memset(info, 0, sizeof *info);
        info->n_texels = (uint32_t)(used_bytes / 4);
        info->n_records = (uint32_t)lay.records.size();
        info->n_internal = lay.n_internal;
        info->n_leaves = lay.n_leaves;
        info->max_depth = lay.max_depth;
    }
    return (long)lay.records.size();
}

// The feedback scheduler's order kernel on a synthetic frame: tile_ticks[n_groups * 4] -> order[n_groups] and the split count the
// kernel leaves behind it (KArgs::split_count). Host arrays, synchronous.
int vrt_test_tile_order(int device, const uint32_t *tile_ticks, uint32_t n_groups, uint32_t wave_slots, uint32_t *order_out, uint32_t *split_out) {
    if (!tile_ticks || !order_out || !split_out || n_groups < 1 || n_groups > 36864) return VRT_E_INVALID;
    void *c = nullptr; (void)c;
    VRT_HIP(c, hipSetDevice(device));
    uint32_t *d_cost = nullptr, *d_order = nullptr;
    VRT_HIP(c, hipMalloc((void **)&d_cost, (size_t)n_groups * 4 * sizeof(uint32_t)));
    VRT_HIP(c, hipMalloc((void **)&d_order, ((size_t)n_groups + 1) * sizeof(uint32_t)));
    VRT_HIP(c, hipMemcpy(d_cost, tile_ticks, (size_t)n_groups * 4 * sizeof(uint32_t), hipMemcpyHostToDevice));
    const size_t lds = (size_t)n_groups * sizeof(uint32_t);
    if (lds > 48 * 1024) VRT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&vrt::tile_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 36864 * 4));
    hipLaunchKernelGGL(vrt::tile_order_kernel, dim3(1), dim3(1024), lds, 0, (const uint4 *)d_cost, n_groups, d_order, wave_slots);
    VRT_HIP(c, hipGetLastError());
    VRT_HIP(c, hipDeviceSynchronize());
    VRT_HIP(c, hipMemcpy(order_out, d_order, (size_t)n_groups * sizeof(uint32_t), hipMemcpyDeviceToHost));
    VRT_HIP(c, hipMemcpy(split_out, d_order + n_groups, sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d_cost);
    (void)hipFree(d_order);
    return VRT_OK;
}

}  // extern "C"
