// vrt_launch_ab.hip -- A/B builds only (make AB=1): the full path tracer as two kernels with rays repacked across waves
// (ab/vrt_bounce.hip.h: deferred-bounce queues, ballot/popcount lane refill). Bit-exact, measured slower than the one-kernel form
// (profiles/r02_b_*: 0.388 against 0.260 ms at 1080p) and therefore not in the shipped library.
#ifdef VRT_AB_VARIANTS
#include "vrt_launch_impl.hip.h"
#include "ab/vrt_bounce.hip.h"

namespace vrt {
namespace launch {
hipError_t trace_split(const KArgs &a, const ViewSet &vs, int grid, int bounce_waves, int refill_below, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    hipError_t e = launch_sched<3, v3::Trav, 8, 64, 5>(a, vs, grid, 0, s, ev0, nullptr);
    if (e != hipSuccess) return e;
    bounce::Args b;
    b.out_rgba = vs.v[0].out_rgba;
    b.refill_below = refill_below;
    if (b.out_rgba) {   // without a colour image there is nothing for the bounce rays to finish
        if (ev1) hipExtLaunchKernelGGL(bounce::bounce_kernel, dim3(bounce_waves), dim3(64), 0, s, nullptr, ev1, 0, a, b);
        else hipLaunchKernelGGL(bounce::bounce_kernel, dim3(bounce_waves), dim3(64), 0, s, a, b);
        return hipGetLastError();
    }
    return ev1 ? hipEventRecord(ev1, s) : hipSuccess;
}
}  // namespace launch
}  // namespace vrt
#endif
