// vrt_bounce.hip.h -- the second half of the full path tracer (shaders/raytracing.comp:435-622) when it runs as two
// kernels: the diffuse bounce rays (comp:596-616) that trace_kernel<3> left in the deferred-bounce queues
// (vrt_full.hip.h defer_bounce) are marched here by PERSISTENT waves whose lanes are refilled from the queue.
//
// Why: a pixel's bounce ray leaves its surface in a random direction of the hemisphere, so the 64 rays of a pixel tile
// take wildly different numbers of DDA steps (a tile's slowest bounce ray takes 2.8 x its mean ray, tests/golden: per-pixel
// fetch counts of the oracle) and the 49 % of the bench frame's pixels that missed have none at all: in the one-kernel
// form 34-37 of a wave's 64 lanes are active on average (profiles/r01_i_pmc_summary.txt). Here rays are not tied to
// pixels any more:
//   * trace_kernel<3> appends the bounce rays of its tile to one of 64 queues (the wave's lanes that defer take
//     consecutive slots with one atomic: ballot + popcount), so the queues hold only rays that exist;
//   * a wave of this kernel keeps per-lane ray state; whenever fewer than kRefillBelow of its lanes still march
//     (__ballot of the active lanes, popcount), the finished lanes shade their result and write their pixel, and every
//     idle lane takes the next ray of the wave's queue (rank among the idle lanes by mbcnt, one atomic per refill);
//   * a wave whose queue is empty moves on to the next queue that still holds rays (lane i looks at queue i, one ballot),
//     so the grid drains all 64 together.
// The per-ray arithmetic is the march of vrt_kernels_v4.hip.h and the depth >= 1 shading of trace_pixel_full, operation
// for operation, so the pixels are the ones the one-kernel form writes (tests/test_gpu_parity.py).
#pragma once
#include "../vrt_full.hip.h"
#include "../vrt_kernels_v4.hip.h"

namespace vrt {
namespace bounce {

constexpr int kRefillBelow = 40;   // refill when fewer lanes than this are still marching (sweep: profiles/r02_bounce_refill.txt)

struct Args {
    uint32_t *out_rgba;            // the view's colour image (full-frame or compact addressing: the records carry offsets)
    int refill_below;              // refill when fewer lanes than this still march (kRefillBelow; tools sweep it)
};

enum : int { kIdle = 0, kActive = 1, kDoneMiss = 2, kDoneHit = 3 };

using T = v4::TravAny;

// The lane's march state: what v4::TravT::march() keeps in registers across its loop.
struct Lane {
    F3 rp, dir, inv, push, dposf;
    v4::Walk w;
    v4::Found cur;
    uint32_t px, py;       // the voxel before `cur` (to_cell4 form)
    uint32_t iof_b;        // the ray's starting medium as a refraction byte
    int steps, axis;
    uint32_t rec;          // index of the ray's record (queue * cap + slot)
};

VRT_DEV I3 dpos_of(F3 dposf) { return I3{dposf.x > 0.0f ? 1 : 0, dposf.y > 0.0f ? 1 : 0, dposf.z > 0.0f ? 1 : 0}; }

// hitMarching's set-up (comp:248-271) for a ray without a prepared first lookup: v4::TravT::march() up to its loop
VRT_DEV void begin(const KArgs &a, const T::Ctx &c, F3 origin, F3 dir, uint32_t iof_b, Lane &l) {
    l.rp = origin;
    const float inv_len = 1.0f / __builtin_sqrtf(dot3(dir, dir));
    dir = scale3(dir, inv_len);
    l.dir = dir;
    l.inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
    l.inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
    l.inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
    l.dposf = F3{dir.x > 0.0f ? 1.0f : 0.0f, dir.y > 0.0f ? 1.0f : 0.0f, dir.z > 0.0f ? 1.0f : 0.0f};
    l.push = F3{sign_c(dir.x) * 0.0001f, sign_c(dir.y) * 0.0001f, sign_c(dir.z) * 0.0001f};
    F3 pf;
    I3 mp;
    T::floor_both(l.rp, pf, mp);
    l.cur.x = 0u; l.cur.y = 85u | (1u << 23); l.cur.plane = F3{0.0f, 0.0f, 0.0f};
    T::reset(l.w);
    const I3 dpos = dpos_of(l.dposf);
    if (T::find(a, c, mp, pf, dpos, l.dposf, l.w, l.cur, false) == v4::kOutside) { l.cur.plane = T::world_planes(a, dpos); T::reset(l.w); }
    l.px = 0u; l.py = 85u | (1u << 23);
    l.iof_b = iof_b;
    l.steps = 0; l.axis = 2;
}

// one iteration of the march loop (v4::TravT::march_loop<false>): returns kActive, kDoneHit or kDoneMiss
VRT_DEV int step(const KArgs &a, const T::Ctx &c, Lane &l) {
    const T::Axis ax = T::dda_step(l.rp, l.dir, l.inv, l.push, l.cur.plane);
    l.axis = ax.x ? 0 : (ax.yz ? 1 : 2);
    F3 pf;
    I3 mp;
    T::floor_both(l.rp, pf, mp);
    const uint32_t cur_m = l.cur.y & 0xffu;
    const uint32_t prev_m = ((l.cur.x >> 24) == 0u || (l.cur.y & (1u << 29)) != 0u) ? l.iof_b : cur_m;
    l.px = l.cur.x; l.py = l.cur.y;
    const int status = T::find(a, c, mp, pf, dpos_of(l.dposf), l.dposf, l.w, l.cur, false);
    const bool hit = status != v4::kOutside && (l.cur.y & 0xffu) != prev_m;
    ++l.steps;
    if (hit) return kDoneHit;
    return (status == v4::kOutside || l.steps >= 1024) ? kDoneMiss : kActive;
}

// what trace_pixel_full does with a popped ray of depth 1 (comp:477-495 for a miss, :497-537 and :573-594 for a hit),
// then the pixel's colour (comp:643)
VRT_DEV void finish(const KArgs &a, const Lane &l, bool hit, uint32_t *out_rgba) {
    const float kPI = 3.14159265359f;
    const float sky[3] = {0.5f, 0.7f, 1.0f};
    const float kSun = 3.0f;
    const size_t plane = (size_t)kDeferQueues * a.defer_cap;
    const float *r = a.defer_rec + l.rec;
    float tc[3] = {r[6 * plane], r[7 * plane], r[8 * plane]};
    float fc[3] = {r[9 * plane], r[10 * plane], r[11 * plane]};
    const float iof = r[12 * plane], weight = r[13 * plane];
    const float mc[3] = {r[14 * plane], r[15 * plane], r[16 * plane]};
    const float md = r[17 * plane];
    const uint32_t out_offset = __float_as_uint(r[18 * plane]);
    if (!hit) {   // depth > 0: the sun-lit sky (distanceInMedium is 0 for a bounce ray that never hit: no absorption)
#pragma unroll
        for (int k = 0; k < 3; ++k) fc[k] = fc[k] + tc[k] * sky[k] * kSun * weight / kPI;
    } else {
        const F3 o{r[0 * plane], r[1 * plane], r[2 * plane]};
        const F3 hp = l.rp;
        const F3 hpw{hp.x / a.voxel_scale, hp.y / a.voxel_scale, hp.z / a.voxel_scale};
        const float dim = 0.0f + len3(sub3(hpw, o)) / a.voxel_scale;
        const I3 mp{(int)__builtin_floorf(hp.x), (int)__builtin_floorf(hp.y), (int)__builtin_floorf(hp.z)};
        Decoded hv = decode_leaf(l.cur.x, v4::word1_of(l.cur.x, l.cur.y));
        Decoded last = decode_leaf(l.px, v4::word1_of(l.px, l.py));
        if (hv.c[3] <= 0.0f) { hv.p[0] = 1.0f; hv.p[1] = 0.0f; hv.p[2] = 0.0f; }
        (void)last.p; (void)iof;   // n1 / n2 only feed rays a depth-1 hit never spawns
        float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sc[k] = hv.c[3] > 0.0f ? hv.c[k] : last.c[k];
        if (dim > 1e-6f && md > 0.0f) full::absorb(tc, md, dim, mc);
        if (mp.x == a.highlighted[0] && mp.y == a.highlighted[1] && mp.z == a.highlighted[2]) {
            sc[0] = 1.0f - sc[0]; sc[1] = 1.0f - sc[1]; sc[2] = 1.0f - sc[2]; sc[3] = 1.0f;
        }
        const float emission = hv.p[1] * 10.0f;
        if (emission > 0.0f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + tc[k] * sc[k] * emission * weight / kPI;
        } else {
            const float amb = fmax_c(1.0f - det_expf(-dim / 512.0f), 0.01f);
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + amb * sc[k] * tc[k] * weight / kPI;
        }
    }
    out_rgba[out_offset] = unorm8(fc[0]) | (unorm8(fc[1]) << 8) | (unorm8(fc[2]) << 16) | (255u << 24);
}

// One wave per workgroup; the grid is sized to fill the chip once (vrt_launch_ab.hip). Every wave reaches the exit: the loop
// ends when no lane marches and all queues have been handed out.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void bounce_kernel(const KArgs a, const Args b) {
    T::Ctx c;
    c.root = a.nodes[0];
    const uint32_t lane_id = threadIdx.x & 63u;
    Lane l;
    l.rp = l.dir = l.inv = l.push = l.dposf = F3{0.0f, 0.0f, 0.0f};
    T::reset(l.w);
    l.cur.x = l.cur.y = 0u; l.cur.plane = F3{0.0f, 0.0f, 0.0f};
    l.px = l.py = 0u; l.iof_b = 85u; l.steps = 0; l.axis = 2; l.rec = 0u;
    int state = kIdle;
    uint32_t queue = blockIdx.x % kDeferQueues;   // wave-uniform: where this wave looks for rays first
    bool exhausted = false;                       // wave-uniform: every queue has been handed out
    const size_t plane = (size_t)kDeferQueues * a.defer_cap;
    for (;;) {
        const uint64_t active = __builtin_amdgcn_ballot_w64(state == kActive);
        const int n_active = __builtin_popcountll(active);
        if (n_active < b.refill_below && (!exhausted || n_active == 0)) {
            if (state >= kDoneMiss) {   // finished lanes: shade, write the pixel
                finish(a, l, state == kDoneHit, b.out_rgba);
                state = kIdle;
            }
            while (!exhausted) {   // idle lanes take the next rays of a queue that still has some
                const uint64_t idle = __builtin_amdgcn_ballot_w64(state == kIdle);
                const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
                if (n_idle == 0u) break;
                // lane i looks at queue i: which queues still hold rays? (a load that bypasses L1: other waves advance these
                // counters; a stale answer only costs one more round)
                const uint32_t handed = __hip_atomic_load(&a.defer_count[(kDeferQueues + lane_id) * kDeferStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t written = a.defer_count[lane_id * kDeferStride];
                const uint64_t avail = __builtin_amdgcn_ballot_w64(handed < written);
                if (avail == 0ull) { exhausted = true; break; }
                const uint64_t rot = queue ? ((avail >> queue) | (avail << (64u - queue))) : avail;   // the first one at or after `queue`
                queue = (queue + (uint32_t)__builtin_ctzll(rot)) % kDeferQueues;
                const uint32_t count = (uint32_t)__builtin_amdgcn_readlane((int)written, (int)queue);
                uint32_t base = 0u;
                if (lane_id == 0u) base = atomicAdd(&a.defer_count[(kDeferQueues + queue) * kDeferStride], n_idle);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (state == kIdle) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                    const uint32_t slot = base + rank;
                    if (slot < count) {
                        l.rec = queue * a.defer_cap + slot;
                        const float *r = a.defer_rec + l.rec;
                        const F3 o{r[0 * plane], r[1 * plane], r[2 * plane]}, d{r[3 * plane], r[4 * plane], r[5 * plane]};
                        begin(a, c, o, d, full::iof_to_byte(r[12 * plane]), l);
                        state = kActive;
                    }
                }
                if (base + n_idle <= count) break;   // every idle lane has a ray now
            }
            if (__builtin_amdgcn_ballot_w64(state == kActive) == 0ull) {
                if (exhausted) break;   // nothing marches and nothing is left to hand out
                continue;
            }
        }
        if (state == kActive) state = step(a, c, l);
    }
}

}  // namespace bounce
}  // namespace vrt
