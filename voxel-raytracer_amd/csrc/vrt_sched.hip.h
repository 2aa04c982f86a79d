// vrt_sched.hip.h -- kernels beside the tracer that exist once (included by vrt_launch_misc.hip only): the order kernel of the
// feedback tile scheduler and the kernarg layout probe.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {

// Feedback scheduling, second half: turns the per-tile ticks of one frame into the group order of the next ones.
// The hardware starts workgroups in index order, and a launch ends when its last-started, slowest workgroups drain; started
// heaviest first, the tail consists of the cheapest tiles instead (longest-processing-time-first list scheduling). A
// group's cost is the maximum over its kGroupTiles tiles; groups are bucketed by cost (256 linear buckets up to the
// frame's maximum) and written out from the heaviest bucket down. One workgroup of 1024 lanes; any permutation is
// correct for the trace kernel, the costs only decide how good it is. group_order has n_groups + 1 words: the last one is
// KArgs::split_count.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint4 *group_ticks, uint32_t n_groups, uint32_t *group_order, uint32_t wave_slots) {
    // group_ticks: the tile_cost words, kGroupTiles per group; the words past the launch's last tile are zero
    static_assert(kGroupTiles == 4, "one 16-byte load per group");
    extern __shared__ uint32_t group_cost[];  // n_groups
    __shared__ uint32_t hist[256], top, n_split;
    __shared__ unsigned long long all_ticks;
    const uint32_t t = threadIdx.x;
    if (t < 256) hist[t] = 0;
    if (t == 0) { top = 0; all_ticks = 0ull; }
    __syncthreads();
    uint32_t m = 0;
    unsigned long long sum = 0ull;
    for (uint32_t base = 0; base < n_groups; base += 8 * 1024) {  // eight loads in flight per lane: one round trip per 8192 groups
        uint4 v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t g = base + u * 1024 + t;
            v[u] = g < n_groups ? group_ticks[g] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t g = base + u * 1024 + t;
            const uint32_t c01 = v[u].x > v[u].y ? v[u].x : v[u].y, c23 = v[u].z > v[u].w ? v[u].z : v[u].w;
            const uint32_t c = c01 > c23 ? c01 : c23;
            if (g < n_groups) group_cost[g] = c;
            m = c > m ? c : m;
            sum += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {  // one atomic per wave, not per lane
        const uint32_t v = (uint32_t)__shfl_xor((int)m, off);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += (unsigned long long)__shfl_xor((long long)sum, off);
    if ((t & 63u) == 0u) { atomicMax(&top, m); atomicAdd(&all_ticks, sum); }
    __syncthreads();
    const uint32_t shift = top >= 256 ? 24 - (uint32_t)__builtin_clz(top) : 0;  // top >> shift <= 255
    for (uint32_t g = t; g < n_groups; g += 1024) atomicAdd(&hist[group_cost[g] >> shift], 1u);
    __syncthreads();
    // hist[b] := first output slot of bucket b, heaviest bucket first: a suffix sum over the 256 counts, done by
    // the first wave alone (lane l owns buckets 4l .. 4l+3) so that it costs one barrier instead of sixteen
    if (t < 64) {
        const uint32_t h0 = hist[4 * t], h1 = hist[4 * t + 1], h2 = hist[4 * t + 2], h3 = hist[4 * t + 3];
        const uint32_t own = h0 + h1 + h2 + h3;
        uint32_t incl = own;  // becomes the sum over lanes >= t
#pragma unroll
        for (uint32_t off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_down((int)incl, off);
            if (t + off < 64) incl += v;
        }
        const uint32_t above = incl - own;  // everything in heavier lanes
        hist[4 * t + 3] = above;
        hist[4 * t + 2] = above + h3;
        hist[4 * t + 1] = above + h3 + h2;
        hist[4 * t] = above + h3 + h2 + h1;
    }
    __syncthreads();
    // KArgs::split_count, kept behind the order: the groups in the buckets from 3/4 of the heaviest bucket up are exactly the first
    // hist[b - 1] of the order (b their lowest bucket). At most kSplitMaxGroups of them (the heaviest come first in the order; a hard
    // "none beyond the limit" made a frame whose count hovers around it fall back to whole tiles every other period), and none unless
    // the frame IS bound by its longest wave: the heaviest tile's ticks against the ticks of all tiles shared out over the chip's wave
    // slots (the room at 1080p: 1.7 times as long, at 4K 0.5 -- there the part-waves' extra instructions would only cost; a frame of
    // equal tiles: 0.2)
    if (t == 0) {
        const uint32_t b = ((top >> shift) * 3u + 3u) / 4u;
        const uint32_t heavy = b == 0u ? n_groups : hist[b - 1u];
        const bool tail_bound = wave_slots != 0u && (unsigned long long)top * wave_slots * 4ull > all_ticks * 3ull;   // top > 3/4 of the even share
        n_split = (top != 0u && tail_bound) ? (heavy < (uint32_t)kSplitMaxGroups ? heavy : (uint32_t)kSplitMaxGroups) : 0u;
    }
    __syncthreads();
    if (t == 0) group_order[n_groups] = n_split;
    for (uint32_t g = t; g < n_groups; g += 1024) group_order[atomicAdd(&hist[group_cost[g] >> shift], 1u)] = g;
}

// late_args() / late_view() assume the kernarg segment holds KArgs at offset 0 and ViewSet behind it at its natural
// alignment. The compiler's layout rules guarantee that for two by-value aggregates, and these keep it from drifting:
static_assert(__is_trivially_copyable(KArgs) && __is_trivially_copyable(ViewSet), "kernel arguments are copied bytewise");
static_assert(alignof(KArgs) <= 8 && alignof(ViewSet) <= 8, "by-value kernel arguments are laid out at their natural alignment (<= 8 here)");
// ... and this probe checks it on the device once per context (vrt_create): every view's late pointers and a few late
// uniforms against the by-value arguments. out[0] = number of mismatches.
__global__ void kernarg_probe_kernel(const KArgs a, const ViewSet vs, uint32_t *out) {
    const LateArgs la = late_args();
    const LateView lv = late_view();
    const View &v = vs.v[blockIdx.y];
    uint32_t bad = 0;
    bad += la->width != a.width || la->height != a.height || la->tex_dim != a.tex_dim || la->compact != a.compact;
    bad += la->light_dir[2] != a.light_dir[2] || la->highlighted[1] != a.highlighted[1] || la->voxel_scale != a.voxel_scale;
    bad += lv->out_rgba != v.out_rgba || lv->out_id != v.out_id || lv->cam_pos[1] != v.cam_pos[1];
    if (threadIdx.x == 0 && bad) atomicAdd(out, bad);
}

}  // namespace vrt
