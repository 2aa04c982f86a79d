// vrt_launch_impl.hip.h -- the instantiated (traversal, LDS prefix, tile width, workgroup size, waves-per-SIMD) combinations of
// trace_kernel, shared by the per-mode launch files (vrt_launch_primary / _shadow / _full .hip: one mode each, so the three
// compile side by side).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "vrt_internal.h"
#include "vrt_launch.h"
#include "vrt_kernels.hip.h"
#include "vrt_kernels_v1.hip.h"
#include "vrt_kernels_wide.hip.h"
#include "vrt_kernels_v4.hip.h"
#include "vrt_full.hip.h"

namespace vrt {
namespace launch {

template <int MODE, class TRAV, int TW, int BLOCK, int WPE, bool PERSIST = false, int SCHED = 0>
// ev0/ev1 (both or neither): events attached to THIS dispatch packet (hipExtLaunchKernel), so their elapsed time is
// the kernel's own begin-to-end time, as a profiler reports it, without the latency of separate event markers
hipError_t launch_one(const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds_bytes, hipStream_t s,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {  // grid.y = a.n_views
    void (*kernel)(const vrt::KArgs, const vrt::ViewSet) = &vrt::trace_kernel<MODE, TRAV, TW, BLOCK, WPE, PERSIST, SCHED>;
    if (lds_bytes > 48 * 1024) {  // above the default dynamic-LDS ceiling: opt in (CDNA4 has 160 KiB per CU). The attribute
        // belongs to the (function, device) pair, so it is set on every such launch (LDS-staging A/B variants only)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
    }
    if (ev0 || ev1)   // either may be null: the two-kernel full path tracer times from the first kernel's start to the second one's end
        hipExtLaunchKernelGGL(kernel, dim3(grid, a.n_views), dim3(BLOCK), lds_bytes, s, ev0, ev1, 0, a, vs);
    else
        hipLaunchKernelGGL(kernel, dim3(grid, a.n_views), dim3(BLOCK), lds_bytes, s, a, vs);
    return hipGetLastError();
}

// The combinations that exist in the feedback-scheduled flavours too (KArgs::group_order / tile_cost choose one).
template <int MODE, class TRAV, int TW, int BLOCK, int WPE>
hipError_t launch_sched(const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    switch ((a.group_order ? 1 : 0) | (a.tile_cost ? 2 : 0)) {
        case 1: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 2>(a, vs, grid, lds, s, ev0, ev1);
        case 3: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 3>(a, vs, grid, lds, s, ev0, ev1);
        default: return launch_one<MODE, TRAV, TW, BLOCK, WPE>(a, vs, grid, lds, s, ev0, ev1);
    }
}

// The instantiated (traversal, LDS prefix, tile width, workgroup size, waves-per-SIMD) combinations.
template <int MODE>
hipError_t launch_mode(const Variant &v, const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds, hipStream_t s,
                       hipEvent_t ev0, hipEvent_t ev1) {
    using V1 = vrt::v1::Trav<false>;
    using V2 = vrt::v2::Trav<false>;
#if VRT_AB
    using V1L = vrt::v1::Trav<true>;
    using V2L = vrt::v2::Trav<true>;
#endif
    using V3 = vrt::v3::Trav;
    using V4 = vrt::v4::Trav;
#if VRT_AB
    if (v.blocks_per_cu > 0) {  // the persistent (grid-stride) form exists for one combination
        if (v.trav == 2 && !v.use_lds && v.tw == 8 && v.block == 256 && v.wpe == 1)
            return launch_one<MODE, V2, 8, 256, 1, true>(a, vs, grid, lds, s, ev0, ev1);
        return hipErrorInvalidValue;
    }
#endif
    const int key = v.trav * 1000000 + (v.use_lds ? 100000 : 0) + v.tw * 1000 + (v.block / 64) * 10 + v.wpe;
    switch (key) {
        // shipped: the default (v4) and the three fallbacks the dispatcher may take
        case 4000000 + 8000 + 10 + 7: return launch_sched<MODE, V4, 8, 64, 7>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 6: return launch_sched<MODE, V3, 8, 64, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 7: return launch_sched<MODE, V3, 8, 64, 7>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 1: return launch_one<MODE, V2, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 1000000 + 8000 + 40 + 1: return launch_one<MODE, V1, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
#if VRT_AB
        case 1100000 + 8000 + 40 + 1: return launch_one<MODE, V1L, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 5: return launch_one<MODE, V2, 8, 256, 5>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 6: return launch_one<MODE, V2, 8, 256, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 8: return launch_one<MODE, V2, 8, 256, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 16000 + 40 + 1: return launch_one<MODE, V2, 16, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 80 + 1: return launch_one<MODE, V2, 8, 512, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 10 + 1: return launch_one<MODE, V2, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 20 + 1: return launch_one<MODE, V2, 8, 128, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2100000 + 8000 + 40 + 1: return launch_one<MODE, V2L, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2100000 + 8000 + 160 + 1: return launch_one<MODE, V2L, 8, 1024, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 1: return launch_one<MODE, V3, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 6: return launch_sched<MODE, V3, 8, 256, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 8: return launch_one<MODE, V3, 8, 256, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 1: return launch_one<MODE, V3, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 8: return launch_one<MODE, V3, 8, 64, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 16000 + 40 + 1: return launch_one<MODE, V3, 16, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 6: return launch_sched<MODE, V4, 8, 64, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 8: return launch_sched<MODE, V4, 8, 64, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 1: return launch_sched<MODE, V4, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace launch
}  // namespace vrt
