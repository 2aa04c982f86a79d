// vrt_display.cpp -- the display pass that consumes the two images in the reference's frame loop (the fullscreen quad drawn by
// src/main.cpp:951-967 with shaders/quad.frag:22-83) and the fused frame call. Host code; the kernel is behind vrt_launch.h.
#include "vrt_internal.h"

#include <cmath>
#include <cstdio>
#include <new>

using namespace vrt_internal;
#include "vrt_launch.h"

extern "C" {

int vrt_denoise(vrt_ctx *c, int width, int height, const void *d_rgba8, const void *d_id_dist, void *d_out_rgba8, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!d_rgba8 || !d_id_dist || !d_out_rgba8 || d_rgba8 == d_out_rgba8) return vrt_fail(c, VRT_E_INVALID, "vrt_denoise: null or aliased buffers");
    VRT_HIP(c, hipSetDevice(c->device));
    vrt::launch::Denoise d{d_rgba8, d_id_dist, d_out_rgba8, width, height, nullptr, nullptr};
    d.rows_path = c->denoise_variant >= 2 ? c->denoise_variant : 0;
    const hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->denoise_variant == 1) {
        VRT_HIP(c, vrt::launch::denoise(d, 1, false, s));
        return VRT_OK;
    }
    // the trace kernel's feedback scheduling, keyed as mode kSchedDenoise: one tile = one workgroup here
    int tiles_x = 0, n_tiles = 0;
    vrt::launch::denoise_tiling(width, height, tiles_x, n_tiles);
    const long groups = ((long)n_tiles + vrt::kGroupTiles - 1) / vrt::kGroupTiles;
    SchedState *st = nullptr;
    if (c->sched_period > 0 && groups >= kSchedMinDenoiseGroups && groups <= kSchedMaxDenoiseGroups)
        st = sched_state(c, s, width, height, 0, 0, 0, kSchedDenoise, (uint32_t)n_tiles, (uint32_t)groups);
    if (!st) {
        VRT_HIP(c, vrt::launch::denoise(d, 0, false, s));
        return VRT_OK;
    }
    const bool measure = measuring_launch(st->launches, c->sched_period);
    d.group_order = st->valid ? st->d_order : nullptr;
    if (measure) {
        d.tile_cost = st->d_cost;
        VRT_HIP(c, hipMemsetAsync(st->d_cost, 0, (size_t)groups * vrt::kGroupTiles * sizeof(uint32_t), s));
    }
    VRT_HIP(c, vrt::launch::denoise(d, 0, true, s));
    ++st->launches;
    if (measure) {
        const int rr = launch_order_kernel(c, st, s);
        if (rr) return rr;
    }
    return VRT_OK;
}

int vrt_denoise_host(vrt_ctx *c, int width, int height, const uint8_t *rgba8, const int32_t *id_dist, uint8_t *out_rgba8) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!rgba8 || !id_dist || !out_rgba8) return vrt_fail(c, VRT_E_INVALID, "vrt_denoise_host: null buffer");
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * (size_t)height;
    r = ensure_scratch(c, px);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(c->d_rgba, rgba8, px * 4, hipMemcpyHostToDevice, c->stream));
    VRT_HIP(c, hipMemcpyAsync(c->d_id, id_dist, px * 8, hipMemcpyHostToDevice, c->stream));
    r = vrt_denoise(c, width, height, c->d_rgba, c->d_id, c->d_shown, nullptr);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(out_rgba8, c->d_shown, px * 4, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_dispatch_frame(vrt_ctx *c, int width, int height, int mode, uint8_t *out_shown_rgba8, uint8_t *out_rgba8,
                       int32_t *out_id_dist) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!out_shown_rgba8) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_frame: null output");
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * (size_t)height;
    r = ensure_scratch(c, px);
    if (r) return r;
    r = enqueue(c, width, height, 0, height, height, 0, 0, mode, c->d_rgba, c->d_id, c->stream);
    if (r) return r;
    r = vrt_denoise(c, width, height, c->d_rgba, c->d_id, c->d_shown, nullptr);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(out_shown_rgba8, c->d_shown, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_rgba8) VRT_HIP(c, hipMemcpyAsync(out_rgba8, c->d_rgba, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_id_dist) VRT_HIP(c, hipMemcpyAsync(out_id_dist, c->d_id, px * 8, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

}  // extern "C"
