// vrt_launch.h -- the seam between the host code of libvrt_hip.so and its device code. Only the vrt_launch_*.hip files include
// kernel headers; everything else launches through these functions.
#pragma once
#include "vrt_internal.h"

namespace vrt {
namespace launch {

// trace_kernel<MODE, ...> for the variant the dispatcher settled on (vrt_launch_primary / _shadow / _full .hip). ev0 / ev1 (both
// or neither): events attached to THIS dispatch packet, so their elapsed time is the kernel's own begin-to-end time.
// hipErrorInvalidValue: the combination is not in this build.
hipError_t trace_primary(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
hipError_t trace_shadow(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
hipError_t trace_full(const Variant &v, const KArgs &a, const ViewSet &vs, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
// The full path tracer as two tile-coherent passes (vrt_full.hip.h bounce_pixel) for scenes the dispatcher has checked: pass 1 = the
// primary + shadow kernel leaving a seed per pixel in a.defer_rec, pass 2 = the diffuse bounce of the seeded pixels. ev0 rides on
// pass 1, ev1 on pass 2 (their elapsed time spans both).
// ... and the same two stages in ONE kernel, the seed in registers (no stack, no seed traffic, no second launch); wpe 5, 6 or 7
hipError_t trace_full_opaque(const KArgs &a, const ViewSet &vs, int grid, int wpe, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
hipError_t trace_full_two_pass(const KArgs &a, const ViewSet &vs, int grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
inline hipError_t trace(int mode, const Variant &v, const KArgs &a, const ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0,
                        hipEvent_t ev1) {
    return mode == VRT_MODE_FULL ? trace_full(v, a, vs, grid, s, ev0, ev1)
                                 : (mode == VRT_MODE_PRIMARY ? trace_primary(v, a, vs, grid, lds, s, ev0, ev1) : trace_shadow(v, a, vs, grid, lds, s, ev0, ev1));
}

// vrt_launch_misc.hip
// tile_order_kernel: the per-tile ticks of one launch -> the group order of the next ones. raise_lds: set the kernel's dynamic-LDS
// ceiling first (once per device; needed above 48 KiB).
// wave_slots: waves the chip holds of the kernel the order is for (KArgs::split_count is 0 unless the heaviest tile outlasts its even share); 0: no split count
hipError_t tile_order(const uint32_t *d_cost, uint32_t n_groups, uint32_t *d_order, uint32_t wave_slots, bool raise_lds, size_t lds_ceiling, hipStream_t s);
// checks on the device that the kernarg segment is laid out as late_args() / late_view() assume; *d_bad += mismatches
hipError_t kernarg_probe(const KArgs &a, const ViewSet &vs, uint32_t *d_bad, hipStream_t s);

// the display pass (vrt_denoise.hip.h). variant 0: 32 x 16-pixel tiles, two pixels per lane (tiles_x / n_tiles: its tiling;
// whole_groups: a 1-D grid of whole scheduling groups that reads group_order / writes tile_cost when given); variant 1 (A/B builds):
// the one-pixel-per-lane kernel of round 1.
struct Denoise {
    const void *rgba, *id;
    void *out;
    int width, height;
    const uint32_t *group_order;
    uint32_t *tile_cost;
    int rows_path = 0;   // denoise::Args::rows_path
};
void denoise_tiling(int width, int height, int &tiles_x, int &n_tiles);
hipError_t denoise(const Denoise &d, int variant, bool whole_groups, hipStream_t s);

#if VRT_AB
// vrt_launch_ab.hip -- the full path tracer as two kernels with cross-wave repacking (ab/vrt_bounce.hip.h): an experiment that lost
// (profiles/r02_b_*), A/B builds only
hipError_t trace_split(const KArgs &a, const ViewSet &vs, int grid, int bounce_waves, int refill_below, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1);
#endif

}  // namespace launch
}  // namespace vrt
