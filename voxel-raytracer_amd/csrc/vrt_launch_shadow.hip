// vrt_launch_shadow.hip -- trace_kernel<MODE 1>: primary rays + one notInShadow() ray per opaque hit (comp:333-377, 587)
#include "vrt_launch_impl.hip.h"

namespace vrt {
namespace launch {
hipError_t trace_shadow(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    return launch_mode<1>(v, a, vs, grid, lds, s, ev0, ev1);
}
}  // namespace launch
}  // namespace vrt
