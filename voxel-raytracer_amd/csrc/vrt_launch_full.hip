// vrt_launch_full.hip -- trace_kernel<MODE 2>: the whole of pathTrace (comp:435-622, vrt_full.hip.h). It exists for the v4 traversal
// (64-lane workgroups, five waves per SIMD: 96 VGPRs and no extra spills measured 8-10 % faster than the unconstrained build), for
// v3 (variant 20) and, as fallbacks for scenes without a wide layout, for the record-array traversals in one shape each.
#include "vrt_launch_impl.hip.h"

namespace vrt {
namespace launch {
hipError_t trace_full(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
#ifndef VRT_FULL_WPE
#define VRT_FULL_WPE 5
#endif
    if (v.trav == 4) return launch_sched<2, v4::TravAny, 8, 64, VRT_FULL_WPE>(a, vs, grid, 0, s, ev0, ev1);
    if (v.trav == 3 && v.block == 64) return launch_sched<2, v3::Trav, 8, 64, 5>(a, vs, grid, 0, s, ev0, ev1);
#if VRT_AB
    if (v.trav == 3) return launch_sched<2, v3::Trav, 8, 256, 5>(a, vs, grid, 0, s, ev0, ev1);
#endif
    if (v.trav == 2) return launch_one<2, v2::Trav<false>, 8, 256, 1>(a, vs, grid, 0, s);
    return launch_one<2, v1::Trav<false>, 8, 256, 1>(a, vs, grid, 0, s);
}

hipError_t trace_full_opaque(const KArgs &a, const ViewSet &vs, int grid, int wpe, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (wpe == 7) return launch_sched<6, v4::Trav, 8, 64, 7>(a, vs, grid, 0, s, ev0, ev1);
    if (wpe == 5) return launch_sched<6, v4::Trav, 8, 64, 5>(a, vs, grid, 0, s, ev0, ev1);
    return launch_sched<6, v4::Trav, 8, 64, 6>(a, vs, grid, 0, s, ev0, ev1);
}

hipError_t trace_full_two_pass(const KArgs &a, const ViewSet &vs, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    hipError_t e = launch_sched<4, v4::Trav, 8, 64, 7>(a, vs, grid, 0, s, ev0, nullptr);
    if (e != hipSuccess) return e;
    KArgs b = a;            // pass 2 starts its tiles in the same order; the tile times that are measured are pass 1's
    b.tile_cost = nullptr;
    return launch_sched<5, v4::TravAny, 8, 64, 7>(b, vs, grid, 0, s, nullptr, ev1);
}

}  // namespace launch
}  // namespace vrt
