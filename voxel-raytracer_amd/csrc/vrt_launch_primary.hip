// vrt_launch_primary.hip -- trace_kernel<MODE 0>: primary rays (shaders/raytracing.comp:624-645 + the primary subset of pathTrace)
#include "vrt_launch_impl.hip.h"

namespace vrt {
namespace launch {
hipError_t trace_primary(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    return launch_mode<0>(v, a, vs, grid, lds, s, ev0, ev1);
}
}  // namespace launch
}  // namespace vrt
