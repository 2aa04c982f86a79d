// vrt_launch_misc.hip -- the kernels beside the tracer: the display pass (vrt_denoise.hip.h), the tile-order kernel of the
// feedback scheduler and the kernarg layout probe (both vrt_common.hip.h).
#include <hip/hip_runtime.h>

#include "vrt_launch.h"
#include "vrt_sched.hip.h"
#include "vrt_denoise.hip.h"

namespace vrt {
namespace launch {

hipError_t tile_order(const uint32_t *d_cost, uint32_t n_groups, uint32_t *d_order, uint32_t wave_slots, bool raise_lds, size_t lds_ceiling, hipStream_t s) {
    const size_t lds = (size_t)n_groups * sizeof(uint32_t);
    if (raise_lds) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ceiling);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), lds, s, (const uint4 *)d_cost, n_groups, d_order, wave_slots);
    return hipGetLastError();
}

hipError_t kernarg_probe(const KArgs &a, const ViewSet &vs, uint32_t *d_bad, hipStream_t s) {
    hipLaunchKernelGGL(kernarg_probe_kernel, dim3(1, kMaxViews), dim3(64), 0, s, a, vs, d_bad);
    return hipGetLastError();
}

void denoise_tiling(int width, int height, int &tiles_x, int &n_tiles) {
    tiles_x = (width + denoise::kTW - 1) / denoise::kTW;
    n_tiles = tiles_x * ((height + 15) / 16);
}

hipError_t denoise(const Denoise &d, int variant, bool whole_groups, hipStream_t s) {
    using namespace vrt::denoise;
    Args a;
    a.rgba = (const uint32_t *)d.rgba;
    a.id = (const int2 *)d.id;
    a.out = (uint32_t *)d.out;
    a.width = d.width;
    a.height = d.height;
    denoise_tiling(d.width, d.height, a.tiles_x, a.n_tiles);
    a.group_order = d.group_order;
    a.tile_cost = d.tile_cost;
    a.rows_path = d.rows_path;
    if (variant == 1) {
#if VRT_AB
        const dim3 grid((unsigned)((d.width + kTile - 1) / kTile), (unsigned)((d.height + kTile - 1) / kTile));
        hipLaunchKernelGGL(denoise_kernel, grid, dim3(kTile, kTile), 0, s, a);
        return hipGetLastError();
#else
        return hipErrorInvalidValue;
#endif
    }
    if (!whole_groups) {
        const dim3 grid((unsigned)a.tiles_x, (unsigned)(a.n_tiles / a.tiles_x));
        hipLaunchKernelGGL((denoise_px_kernel<2, 16>), grid, dim3(kTW / 2, 16), 0, s, a);
    } else {
        const long groups = ((long)a.n_tiles + kGroupTiles - 1) / kGroupTiles;
        hipLaunchKernelGGL((denoise_px_kernel<2, 16, true>), dim3((unsigned)(groups * kGroupTiles)), dim3(kTW / 2, 16), 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace launch
}  // namespace vrt
