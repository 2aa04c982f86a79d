// vrt_denoise.hip.h -- the display pass that follows the trace in the reference's frame loop:
// shaders/quad.frag:22-83, an ID-aware box blur. For a pixel whose voxelID is non-zero, every
// pixel of the (2R+1)^2 window around it that carries the SAME voxelID contributes its colour;
// R = clamp(int(200 / sqrt(max(1, dist))), 1, 20), so up to 41 x 41 = 1681 taps per pixel.
//
// gfx950 form: a 16x16-pixel workgroup stages its tile plus a 20-pixel halo (56 x 56 pixels) in LDS
// once -- voxel IDs as int32 and the colours already converted to the floats the shader's sampler
// returns (byte / 255.0f, one correctly rounded division per staged pixel instead of one per tap) --
// 50 KB per workgroup, three workgroups per CU. Every lane then walks its own window in the
// shader's order (y outer, x inner) so the fp32 sums round exactly as the reference's do; window
// rows/columns that fall outside the image are skipped by clamping the loop bounds, which visits
// the surviving taps in the same order. One ds_read for the ID and, on a match, one 12-byte read
// for the colour per tap; global memory is touched only by the staging loads and the final store.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace denoise {

constexpr int kTile = 16;
constexpr int kMaxR = 20;
constexpr int kSpan = kTile + 2 * kMaxR;  // 56

struct Args {
    const uint32_t *rgba;  // packed rgba8, W*H
    const int2 *id;        // (voxelID, dist), W*H
    uint32_t *out;         // packed rgba8, W*H
    int width, height;
};

__global__ __launch_bounds__(kTile *kTile) void denoise_kernel(const Args a) {
    __shared__ int s_id[kSpan * kSpan];
    __shared__ float s_col[kSpan * kSpan * 3];
    const int tx0 = blockIdx.x * kTile - kMaxR, ty0 = blockIdx.y * kTile - kMaxR;
    const int tid = threadIdx.y * kTile + threadIdx.x;
    for (int i = tid; i < kSpan * kSpan; i += kTile * kTile) {
        const int lx = i % kSpan, ly = i / kSpan;
        const int gx = tx0 + lx, gy = ty0 + ly;
        int vid = 0;
        uint32_t c = 0u;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t g = (size_t)gy * (size_t)a.width + (size_t)gx;
            vid = a.id[g].x;
            c = a.rgba[g];
        }
        s_id[i] = vid;
        s_col[3 * i + 0] = (float)(c & 0xffu) / 255.0f;
        s_col[3 * i + 1] = (float)((c >> 8) & 0xffu) / 255.0f;
        s_col[3 * i + 2] = (float)((c >> 16) & 0xffu) / 255.0f;
    }
    __syncthreads();
    const int px = blockIdx.x * kTile + threadIdx.x, py = blockIdx.y * kTile + threadIdx.y;
    if (px >= a.width || py >= a.height) return;
    const size_t p = (size_t)py * (size_t)a.width + (size_t)px;
    const int2 center = a.id[p];
    if (center.x == 0) {  // quad.frag:36-39: sky / no first-hit id: pass the colour through
        a.out[p] = a.rgba[p];
        return;
    }
    const float radius_f = 200.0f / __builtin_sqrtf((float)(center.y > 1 ? center.y : 1));  // :45
    int R = (int)radius_f;
    R = R < 1 ? 1 : (R > kMaxR ? kMaxR : R);  // :48
    // window clipped to the image (:60-63); same visiting order for the taps that remain
    const int y_lo = -R < -py ? -py : -R, y_hi = R > a.height - 1 - py ? a.height - 1 - py : R;
    const int x_lo = -R < -px ? -px : -R, x_hi = R > a.width - 1 - px ? a.width - 1 - px : R;
    const int cx = threadIdx.x + kMaxR, cy = threadIdx.y + kMaxR;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
    int count = 0;
    for (int y = y_lo; y <= y_hi; ++y) {
        const int row = (cy + y) * kSpan + cx;
        for (int x = x_lo; x <= x_hi; ++x) {
            const int i = row + x;
            if (s_id[i] == center.x) {  // :67-73
                s0 = s0 + s_col[3 * i + 0];
                s1 = s1 + s_col[3 * i + 1];
                s2 = s2 + s_col[3 * i + 2];
                ++count;
            }
        }
    }
    const float d = fmax_c((float)count, 1.0f);  // count <= 1681: the float the shader accumulates, exactly
    a.out[p] = unorm8(s0 / d) | (unorm8(s1 / d) << 8) | (unorm8(s2 / d) << 16) | (255u << 24);
}

}  // namespace denoise
}  // namespace vrt
