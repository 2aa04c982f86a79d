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
    // feedback scheduling (SCHED flavour of denoise_px_kernel; the trace kernel's scheme, vrt_common.hip.h): tiles are
    // numbered row-major, a group is kGroupTiles consecutive tiles; either pointer may be null
    int tiles_x, n_tiles;
    const uint32_t *group_order;
    uint32_t *tile_cost;   // atomicMax of the waves' clock ticks; zeroed by the host before a measuring launch
    int rows_path;         // 0: per wave, the cheaper of the two walks below; 2: always the wave's common rows (rows_static); 3: always own boxes (rows_own_box)
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


// ---------------------------------------------------------------------------------------------------
// PX pixels per lane. The one-pixel kernel above reads 16 LDS bytes per tap per pixel and is bound by
// LDS bandwidth and by waiting on it. Here a lane owns PX horizontally adjacent pixels and walks the
// UNION of their windows, x = -R .. R+PX-1 relative to its first pixel: one ds_read_b128 ({r, g, b, id})
// serves up to PX windows, and every pixel still receives its own taps in the shader's order (y outer,
// x inner), so each fp32 sum rounds as before.
//   * the add is an fma with a 0/1 mask (s = fma(m, c, s)): adding +0 leaves a non-negative sum
//     unchanged, so the masked form is bit-identical to the shader's conditional add; (r,g) and
//     (b,count) go through packed fp32 fma: compare, select, two v_pk_fma_f32 per tap and pixel;
//   * taps outside the image are staged with id 0 and can never equal a summed pixel's id (pixels
//     with id 0 pass through), which reproduces quad.frag's bounds test (:60-63) without clipping
//     the loops;
//   * rows are straight-line code specialised on the wave's largest radius RM and on the spread DELTA of
//     its radii (rows_static): compile-time LDS offsets, no scalar address arithmetic, reads hoisted
//     ahead of use; only the DELTA taps at either end of a pixel's span take a per-pixel range test.
//     DELTA 0 (one radius; the common case, R moves slowly across the image) and 1 have their own
//     instances, any larger spread runs the instance that tests every tap.
// Tile 32 x 16 pixels (+20 halo = 72 x 56 taps, 70 KB LDS with the padded row stride, two workgroups per CU).
// Within a row the tap columns are stored column-mod-PX major (position = (col % PX) * (72 / PX) + col / PX) so
// that the lanes of a row, which read columns PX apart, hit consecutive 16-byte slots.
// The shipped instance is PX = 2: 16 x 16 lanes = four waves per workgroup, one per SIMD, so two resident
// workgroups always put two waves on every SIMD. PX = 4 halves the LDS reads and needs 4.25 instead of 4.5
// vector instructions per tap and pixel, but its two-wave workgroups land on SIMDs at the dispatcher's whim and
// measured 30% slower on rendered frames. Measured on MI355X (tools/denoise_time.py, 1080p dragon / 4K nature /
// every pixel summed at radius 20): 0.34 / 0.18 / 0.49 ms against 0.68 / 0.53 / 1.76 ms for the one-pixel kernel.
// What bounds it: v_pk_fma_f32 issues in ~6 cycles, not 4 (tools/micro/valu_rate.hip), which puts a tap-and-pixel
// at ~21 cycles of one SIMD; the dense frame runs within 10% of that, rendered frames lose the rest to the spread
// of per-tile work (sky tiles are free, radius-20 tiles take ~50 us a wave).
// Also measured and not kept (round 2): one comparison and mask per tap shared by a lane's two pixels in waves whose summed
// pixels all carry one id. 17% faster where every wave is like that (the dense frame: 0.405 ms against 0.49), 0-1.5% on
// rendered frames (dragon 1080p 0.341 / 0.281 scheduled against 0.342 / 0.285): at these poses a voxel face is ~13 pixels
// wide and a wave's 32 x 4 pixels nearly always hold several ids; it doubled the one-radius instances and spilled.
constexpr int kTW = 32;                  // tile width; the height TH is a kernel parameter (16 by default)
constexpr int kSpanX = kTW + 2 * kMaxR;  // 72
constexpr int kFull = kMaxR;             // DELTA value of the instance that range-tests every tap
constexpr int kSeg = 4;                  // a row's taps are walked in segments of this many columns; a segment no lane needs is skipped (2, 3, 7, 10 measured slower)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int PX>
struct Acc {
    f2 rg[PX];
    f2 bc[PX];  // (blue sum, tap count)
};

// LDS row stride in 16-byte slots. ds_read_b128 is served 16 lanes at a time, lanes {0-3,12-15,20-27} first
// (MI355X guide, LDS table): with 16 lanes per pixel row (PX = 2) that group spans two rows and is conflict-free
// only when the row stride is a multiple of 256 B; with 8 lanes per row (PX = 4) it spans four rows and the
// natural 72-slot stride (1152 B = 128 mod 256) already spreads them over all banks.
template <int PX>
constexpr int kStride = PX == 2 ? 80 : kSpanX;

template <int PX>
__device__ __forceinline__ int slot(const int col) {
    return (col % PX) * (kSpanX / PX) + col / PX;
}

#ifndef VRT_DENOISE_TAP
#define VRT_DENOISE_TAP 0
#endif
#ifndef VRT_DENOISE_MASKED_TAP   // a 0/1 mask multiplied in: compare, select, then the four sums
// How the four sums are issued (same values in every form: an fma with a 0/1 multiplier IS the conditional add, and count + m is
// fma(m, 1, count)). A gfx950 SIMD issues one vector instruction per ~2.3 cycles whatever its kind and runs the half-rate kinds
// (v_cmp, v_cndmask, every v_pk_*_f32) on a second pipe that needs ~4.3 cycles for each (profiles/r03_valu_rate.txt), so a tap
// costs max(2.3 x instructions, 4.3 x half-rate instructions):
//   0  v_cmp, v_cndmask, 2 x v_pk_fma_f32                      (round 1-2)   4 instructions, 4 half-rate: 17.2
//   1  v_cmp, v_cndmask, v_pk_fma_f32 (r, g), v_fma_f32 (b), v_add_f32 (count)  5 instructions, 3 half-rate: 12.9
//   2  v_cmp, v_cndmask, 3 x v_fma_f32, v_add_f32                           6 instructions, 2 half-rate: 13.8
__device__ __forceinline__ void tap(const f4 rec, const int cid, f2 &rg, f2 &bc, const bool in_range = true) {
    const float m = (__float_as_int(rec.w) == cid && in_range) ? 1.0f : 0.0f;
#if VRT_DENOISE_TAP == 0
    const f2 mm = {m, m};
    const f2 c01 = {rec.x, rec.y};
    const f2 c2 = {rec.z, 1.0f};
    rg = __builtin_elementwise_fma(mm, c01, rg);
    bc = __builtin_elementwise_fma(mm, c2, bc);
#elif VRT_DENOISE_TAP == 1
    const f2 mm = {m, m};
    const f2 c01 = {rec.x, rec.y};
    rg = __builtin_elementwise_fma(mm, c01, rg);
    bc.x = __builtin_fmaf(m, rec.z, bc.x);
    bc.y = bc.y + m;
#else
    rg.x = __builtin_fmaf(m, rec.x, rg.x);
    rg.y = __builtin_fmaf(m, rec.y, rg.y);
    bc.x = __builtin_fmaf(m, rec.z, bc.x);
    bc.y = bc.y + m;
#endif
}
#else
// Tried in round 2 and measured SLOWER (kept for the record, -DVRT_DENOISE_MASKED_TAP): the conditional add as what it
// is -- the comparison's lane mask becomes exec for two packed adds, which saves the v_cndmask (as dear as a packed add,
// profiles/r02_valu_rate.txt) and turns the fmas into adds: 13.2 instead of 17.6 ticks of vector issue per tap and
// pixel. Same pixels, but 1080p dragon 0.463 ms against 0.342 (0.449 / 0.285 with feedback scheduling), monu9 720p
// 0.222 / 0.190, nature 4K 0.230 / 0.178 (profiles/r02_f_denoise_masked_tap.txt): every tap now hangs on a scalar ->
// vector hand-over of exec, and this kernel runs two waves per SIMD (70 KB of LDS per workgroup) -- too few to hide it.
// The mask-multiply form is a pure stream of independent vector instructions, which is what so few waves need.
__device__ __forceinline__ void tap(const f4 rec, const int cid, f2 &rg, f2 &bc, const bool in_range = true) {
    // two ballots and a scalar AND: the ballot of `a && b` is lowered through a select and a second compare
    const uint64_t m = __builtin_amdgcn_ballot_w64(__float_as_int(rec.w) == cid) & __builtin_amdgcn_ballot_w64(in_range);
    const f2 c01 = {rec.x, rec.y};
    const f2 c2 = {rec.z, 1.0f};
    uint64_t saved;
    // not volatile: exec is back to what it was when the statement ends, and a volatile statement keeps the row's LDS reads
    // from being hoisted over it
    asm("s_and_saveexec_b64 %[saved], %[m]\n\t"
                 "v_pk_add_f32 %[rg], %[rg], %[c01]\n\t"
                 "v_pk_add_f32 %[bc], %[bc], %[c2]\n\t"
                 "s_mov_b64 exec, %[saved]"
                 : [rg] "+v"(rg), [bc] "+v"(bc), [saved] "=&s"(saved)
                 : [m] "s"(m), [c01] "v"(c01), [c2] "v"(c2)
                 : "scc");
}
#endif

// RM is the largest radius among the wave's summed pixels, DELTA (wave-uniform) at least RM - the smallest.
// Tap u of a row (x = u - RM relative to pixel 0) can only matter to pixel k when k <= u <= k + 2RM, and it
// lies in EVERY pixel's window, whatever that pixel's own radius, when k + DELTA <= u <= k + 2RM - DELTA and
// the row is one of the middle 2(RM-DELTA)+1: those taps need no test. Pixel k takes any other tap when
// |u - RM - k| <= R[k] and |y| <= R[k].
// `row` is the lane's first window row (y = -RM) at the lane's column base. Each row is straight-line code; the
// compiler hoists the row's LDS reads ahead of their use. (Cutting the row into register double-buffered chunks
// pinned by empty asm statements was measured: 4% faster on a frame where every wave has one radius, 10-25% slower
// on rendered frames, so the rows are left to the scheduler.)
// y_first..y_last (wave-uniform, inside [-RM, RM]): the window rows that can hold a tap of one of the wave's ids at all (tile():
// rows outside it add +0 to every sum and are skipped).
// SEGMENTED: a row's taps are walked in segments of kSeg columns and the segments in which no lane has a tap of its ids are skipped
// (seg_mask, wave-uniform); the other instance is the row as ONE piece of straight-line code, for waves that need every column
// (a close-up, one face filling the window): the segment branches keep the compiler from hoisting a whole row's LDS reads, which
// costs such a wave 14 %.
template <int PX, int RM, int DELTA, bool SEGMENTED>
__device__ __forceinline__ void rows_static(const f4 *row, const int (&cid)[PX], const int (&R)[PX], Acc<PX> &acc, const int y_first, const int y_last,
                                            const uint32_t seg_mask) {
    constexpr int D = DELTA < RM ? DELTA : RM;
    unsigned extent[PX];
    int inner_first[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        extent[k] = (unsigned)(2 * R[k]);
        inner_first[k] = RM - R[k] + k;
    }
    row += (y_first + RM) * kStride<PX>;
#pragma unroll 1
    for (int y = y_first; y <= y_last; ++y, row += kStride<PX>) {
        const bool middle = D == 0 || (D < RM && y >= -(RM - D) && y <= RM - D);  // wave-uniform
        constexpr int kSegW = SEGMENTED ? kSeg : 2 * RM + PX;   // unsegmented: one segment holds the row
        constexpr int kSegs = (2 * RM + PX + kSegW - 1) / kSegW;
        if (middle) {
#pragma unroll
            for (int sg = 0; sg < kSegs; ++sg) {
                if (SEGMENTED && !((seg_mask >> sg) & 1u)) continue;   // wave-uniform: no lane has a tap of its ids in these columns
#pragma unroll
                for (int u = sg * kSegW; u < (sg + 1) * kSegW && u < 2 * RM + PX; ++u) {
                    const f4 t = row[slot<PX>(kMaxR - RM + u)];
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        if (u < k || u > k + 2 * RM) continue;
                        if (u >= k + D && u <= k + 2 * RM - D)
                            tap(t, cid[k], acc.rg[k], acc.bc[k]);
                        else
                            tap(t, cid[k], acc.rg[k], acc.bc[k], (unsigned)(u - inner_first[k]) <= extent[k]);
                    }
                }
            }
        } else {
            int first[PX];  // a row outside pixel k's window gets a lower bound no tap reaches
#pragma unroll
            for (int k = 0; k < PX; ++k) first[k] = (y >= -R[k] && y <= R[k]) ? inner_first[k] : 4 * kMaxR;
#pragma unroll
            for (int sg = 0; sg < kSegs; ++sg) {
                if (SEGMENTED && !((seg_mask >> sg) & 1u)) continue;
#pragma unroll
                for (int u = sg * kSegW; u < (sg + 1) * kSegW && u < 2 * RM + PX; ++u) {
                    const f4 t = row[slot<PX>(kMaxR - RM + u)];
#pragma unroll
                    for (int k = 0; k < PX; ++k) {
                        if (u < k || u > k + 2 * RM) continue;
                        tap(t, cid[k], acc.rg[k], acc.bc[k], (unsigned)(u - first[k]) <= extent[k]);
                    }
                }
            }
        }
    }
}

template <int PX, int DELTA>
__device__ __forceinline__ void rows_dispatch(const f4 *row, const int rm, const int (&cid)[PX], const int (&R)[PX], Acc<PX> &acc, const int y_first,
                                              const int y_last, const uint32_t seg_mask) {
    const uint32_t every = (1u << ((2 * rm + PX + kSeg - 1) / kSeg)) - 1u;   // all of the row's segments
    if (seg_mask != every) {
        switch (rm) {
#define VRT_DENOISE_CASE(r) \
    case r:                 \
        rows_static<PX, r, DELTA, true>(row, cid, R, acc, y_first, y_last, seg_mask); \
        break;
            VRT_DENOISE_CASE(1) VRT_DENOISE_CASE(2) VRT_DENOISE_CASE(3) VRT_DENOISE_CASE(4) VRT_DENOISE_CASE(5)
            VRT_DENOISE_CASE(6) VRT_DENOISE_CASE(7) VRT_DENOISE_CASE(8) VRT_DENOISE_CASE(9) VRT_DENOISE_CASE(10)
            VRT_DENOISE_CASE(11) VRT_DENOISE_CASE(12) VRT_DENOISE_CASE(13) VRT_DENOISE_CASE(14) VRT_DENOISE_CASE(15)
            VRT_DENOISE_CASE(16) VRT_DENOISE_CASE(17) VRT_DENOISE_CASE(18) VRT_DENOISE_CASE(19) VRT_DENOISE_CASE(20)
#undef VRT_DENOISE_CASE
        }
        return;
    }
    switch (rm) {
#define VRT_DENOISE_CASE(r) \
    case r:                 \
        rows_static<PX, r, DELTA, false>(row, cid, R, acc, y_first, y_last, seg_mask); \
        break;
        VRT_DENOISE_CASE(1) VRT_DENOISE_CASE(2) VRT_DENOISE_CASE(3) VRT_DENOISE_CASE(4) VRT_DENOISE_CASE(5)
        VRT_DENOISE_CASE(6) VRT_DENOISE_CASE(7) VRT_DENOISE_CASE(8) VRT_DENOISE_CASE(9) VRT_DENOISE_CASE(10)
        VRT_DENOISE_CASE(11) VRT_DENOISE_CASE(12) VRT_DENOISE_CASE(13) VRT_DENOISE_CASE(14) VRT_DENOISE_CASE(15)
        VRT_DENOISE_CASE(16) VRT_DENOISE_CASE(17) VRT_DENOISE_CASE(18) VRT_DENOISE_CASE(19) VRT_DENOISE_CASE(20)
#undef VRT_DENOISE_CASE
    }
}

// The other walk: every pixel over its OWN box. rows_static walks, for all 64 lanes alike, the rows and column segments that ANY
// lane of the wave needs -- relative to the lane, so a wave whose pixels sit at all positions inside faces a dozen pixels across
// needs about twice a face's extent each way. Here a pixel walks the box [cx0, cx0 + w) x [ry0, ry0 + h) of staged taps its id
// occurs in (the id's first / last staged row and column from the tile's table, cut to the pixel's window), from a per-lane LDS
// address; the loops run to the largest h and w among the wave's lanes (hmax, wmax: wave-uniform) and a lane masks what lies
// beyond its own box (such a tap could still carry the id where the box was cut by the window, not by the id's extent). Taps
// are taken row by row, columns ascending: the shader's order for the taps that carry the id, and the others add +0.
// Five vector instructions per tap and pixel (two compares, a select, two packed fmas) and an LDS read of its own, against
// four and half a read in rows_static -- but a 13 x 13 box instead of a 27 x 29 union: where faces are small against the window
// (radius 15-20 at the distance of the bench poses) that is a third of the instructions. Waves whose faces fill the window
// (close-ups) keep rows_static: tile() prices both walks per wave. (Measured and not kept: the next trip's reads issued before this
// trip's sums, 20 % slower -- 64 more live registers at two waves per SIMD; two or eight pairs per trip instead of four, 0-2 % slower.)
// Column c of a staged row sits in slot (c % 2) * 36 + c / 2: columns c, c + 2, c + 4 ... are consecutive slots, so the walk
// alternates between two pointers (columns of cx0's parity, and the others).
// The lane's PX pixels are walked at once: one loop nest to the largest box among all of them, PX independent chains of sums and PX times
// the reads in flight per trip (the pass runs two waves per SIMD: what hides an LDS read's latency is the lane's own other work; one
// pixel after the other measured 6 % slower). Eight taps per pixel and trip, their reads issued together: reads past a box are masked;
// past a row's taps they find the row's padding or the next row, staged taps or zeros: finite either way (tile() zeroes the padding).
typedef const f4 __attribute__((address_space(3))) *LdsF4;   // a pointer typed as LDS: a function that is not inlined must not depend on the compiler proving it (a flat pointer is read with flat loads)
template <int PX>
struct Boxes {   // per lane: its pixels' ids and the boxes they occur in (staged rows / columns; h = w = 0: a pixel that is not summed)
    int cid[PX], ry0[PX], h[PX], cx0[PX], w[PX];
};
// Not inlined, by value in and out: rows_static's instances fill the register file by design (a whole row's reads hoisted), and with this
// walk inlined beside them every one of them spilled a dozen registers (the common-rows walk of a close-up frame: 0.51 -> 0.92 ms).
template <int PX>
__device__ __attribute__((noinline)) Acc<PX> rows_own_box(const LdsF4 s_rec, const Boxes<PX> b, const int hmax_, const int wmax_) {
    // arguments of a function that is not inlined arrive in vector registers: the loop bounds are wave-uniform, say so
    const int hmax = __builtin_amdgcn_readfirstlane(hmax_), wmax = __builtin_amdgcn_readfirstlane(wmax_);
    const int (&cid)[PX] = b.cid, (&ry0)[PX] = b.ry0, (&h)[PX] = b.h, (&cx0)[PX] = b.cx0, (&w)[PX] = b.w;
    Acc<PX> acc;
#pragma unroll
    for (int k = 0; k < PX; ++k) { acc.rg[k] = f2{0.0f, 0.0f}; acc.bc[k] = f2{0.0f, 0.0f}; }
    int ia[PX], ib[PX];
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        ia[k] = ry0[k] * kStride<PX> + slot<PX>(cx0[k]);
        ib[k] = ry0[k] * kStride<PX> + slot<PX>(cx0[k] + 1);
    }
    const int pairs = (wmax + 1) >> 1;   // wave-uniform
#pragma unroll 1
    for (int dy = 0; dy < hmax; ++dy) {
        unsigned wv[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) wv[k] = dy < h[k] ? (unsigned)w[k] : 0u;
#pragma unroll 1
        for (int j = 0; j < pairs; j += 4) {
            f4 ta[PX][4], tb[PX][4];
#pragma unroll
            for (int k = 0; k < PX; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) { ta[k][i] = s_rec[ia[k] + j + i]; tb[k][i] = s_rec[ib[k] + j + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    tap(ta[k][i], cid[k], acc.rg[k], acc.bc[k], (unsigned)(2 * (j + i)) < wv[k]);
                    tap(tb[k][i], cid[k], acc.rg[k], acc.bc[k], (unsigned)(2 * (j + i) + 1) < wv[k]);
                }
        }
#pragma unroll
        for (int k = 0; k < PX; ++k)
            if (dy + 1 < h[k]) { ia[k] += kStride<PX>; ib[k] += kStride<PX>; }
    }
    return acc;
}

// One tile (bx, by) by one workgroup of (32 / PX) x TH lanes. Ends with every lane
// past its last LDS read of this tile, but not synchronised.
// Which rows of the staged window hold a given id at all. A pixel's sum only takes taps that carry its own voxelID -- the pixels of
// ONE voxel face -- and at the distances where the radius is large a face is a dozen pixels across while the window is 41 x 41:
// most of a window's rows hold no tap of the id, and a row without one adds +0 to every sum, which leaves it unchanged (the sums
// are non-negative). So the tile keeps, for every id one of its OWN pixels carries, the first and last staged row that id occurs
// in (a 128-slot open-addressing table in LDS: ids claimed with atomicCAS before staging, rows folded in with atomicMin / atomicMax
// while staging), and a wave walks only the rows between the lowest first and the highest last row of its lanes' ids. An id the
// table has no room for switches the tile back to whole windows. Same taps in the same order for every sum: same bits.
constexpr int kIdSlots = 128;
struct IdRows {
    int id[kIdSlots];       // 0 = free (a pixel with id 0 is never summed)
    int lo[kIdSlots], hi[kIdSlots];      // first / last staged row the id occurs in
    int xlo[kIdSlots], xhi[kIdSlots];    // first / last staged column
    int overflow;
};

__device__ __forceinline__ uint32_t id_slot(const int id) { return ((uint32_t)id * 2654435761u) >> 25; }   // 7 bits

template <int PX, int TH>
__device__ __forceinline__ void tile(const Args &a, const int bx, const int by, f4 *s_rec, IdRows *s_ids) {
    constexpr int kTH = TH, kSpanY = TH + 2 * kMaxR;
    constexpr int kLanesX = kTW / PX, kThreads = kLanesX * kTH, kTaps = kSpanX * kSpanY;
    static_assert(kSpanX % PX == 0, "column swizzle");
    const int tid = threadIdx.y * kLanesX + threadIdx.x;
    const int px0 = bx * kTW + threadIdx.x * PX, py = by * kTH + threadIdx.y;
#ifdef VRT_DENOISE_PHASE   // experiment builds (tools/denoise_phases.sh): the ticks of ONE phase of the tile's first wave, left in the tile's first pixel
    unsigned long long ph_t = __builtin_readcyclecounter();
    uint32_t ph_ticks = 0u;
#define VRT_PH(k) do { const unsigned long long ph_n = __builtin_readcyclecounter(); if (VRT_DENOISE_PHASE == (k)) ph_ticks += (uint32_t)(ph_n - ph_t); ph_t = ph_n; } while (0)
#else
#define VRT_PH(k) do { } while (0)
#endif
    int cid[PX], R[PX];
    int r_hi = 0, r_lo = kMaxR + 1;
    for (int i = tid; i < kIdSlots; i += kThreads) {
        s_ids->id[i] = 0;
        s_ids->lo[i] = s_ids->xlo[i] = 0x7fffffff;
        s_ids->hi[i] = s_ids->xhi[i] = -0x7fffffff;
    }
    if (tid == 0) s_ids->overflow = 0;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const int px = px0 + k;
        cid[k] = 0;
        R[k] = 1;
        if (px < a.width && py < a.height) {
            const int2 center = a.id[(size_t)py * (size_t)a.width + (size_t)px];
            cid[k] = center.x;
            const float radius_f = 200.0f / __builtin_sqrtf((float)(center.y > 1 ? center.y : 1));  // quad.frag:45
            const int r = (int)radius_f;
            R[k] = r < 1 ? 1 : (r > kMaxR ? kMaxR : r);                                           // :48
        }
        if (cid[k] != 0) {
            r_hi = R[k] > r_hi ? R[k] : r_hi;
            r_lo = R[k] < r_lo ? R[k] : r_lo;
        }
    }
    VRT_PH(1);   // table init, own pixels loaded, radii
    // a tile of sky (quad.frag:36-39 for every pixel) copies its colours and never stages a window
    if (!__syncthreads_or(r_hi)) {
        if (py < a.height) {
#pragma unroll
            for (int k = 0; k < PX; ++k)
                if (px0 + k < a.width) {
                    const size_t p = (size_t)py * (size_t)a.width + (size_t)(px0 + k);
                    a.out[p] = a.rgba[p];
                }
        }
        return;
    }
#ifdef VRT_DENOISE_POISON   // experiment builds: every LDS word the pass does not write itself reads as NaN, so a sum that depended on one would show
    for (int i = tid; i < kStride<PX> * (kSpanY + 1); i += kThreads) s_rec[i] = f4{__uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u)};
    __syncthreads();
#endif
    // the ids of the tile's own pixels claim their slots (the barrier above ordered the table's initialisation before this)
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        if (cid[k] == 0) continue;
        uint32_t sl = id_slot(cid[k]);
        int tries = 0;
        for (; tries < kIdSlots; ++tries, sl = (sl + 1u) & (kIdSlots - 1)) {
            const int was = atomicCAS(&s_ids->id[sl], 0, cid[k]);
            if (was == 0 || was == cid[k]) break;
        }
        if (tries == kIdSlots) s_ids->overflow = 1;
    }
    __syncthreads();
    VRT_PH(2);   // sky vote, claims, barrier
    // stage the 72 x (TH + 40) window: loads of a whole batch are issued before the first is consumed. (All sixteen of a thread's
    // loads requested up front, before its own pixels' -- one round trip per tile instead of three -- measured no faster on the
    // dragon frame and 30 % slower on the 4K frame, whose sky tiles then load their windows for nothing.)
    const int tx0 = bx * kTW - kMaxR, ty0 = by * kTH - kMaxR;
    constexpr int kBatch = 8;
    constexpr int kStageIters = (kTaps + kThreads * kBatch - 1) / (kThreads * kBatch) * kBatch;
    // rows_own_box reads up to 23 slots past a lane's box and multiplies what it finds by a zero mask: everything such a read can
    // reach must hold finite floats -- the eight padding slots behind each row's 72 taps and one row behind the last
    constexpr int kPad = kStride<PX> - kSpanX;   // 8 with two pixels per lane
    for (int i = tid; i < kSpanY * kPad + kStride<PX>; i += kThreads) {
        int at = kSpanY * kStride<PX> + (i - kSpanY * kPad);            // the row behind the window
        if constexpr (kPad > 0) {
            if (i < kSpanY * kPad) at = (i / kPad) * kStride<PX> + kSpanX + i % kPad;
        }
        s_rec[at] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    VRT_PH(31);   // padding zeroed
    // a thread's taps mostly carry one id after another (its taps lie 3.5 rows apart in one column band): runs of one id are folded
    // in registers and reach the table once per run -- on a close-up where one face fills the tile that is two atomics per thread
    // instead of two per tap on ONE slot
    int run_id = 0, run_slot = -1, run_lo = 0, run_hi = 0, run_xlo = 0, run_xhi = 0;
    const auto flush_run = [&]() {
        if (run_slot >= 0) {
            atomicMin(&s_ids->lo[run_slot], run_lo); atomicMax(&s_ids->hi[run_slot], run_hi);
            atomicMin(&s_ids->xlo[run_slot], run_xlo); atomicMax(&s_ids->xhi[run_slot], run_xhi);
        }
    };
    const auto stage_tap = [&](const int vidj, const uint32_t colj, const int lx, const int ly) {
        f4 rec;
        rec.x = unorm_of((float)(colj & 0xffu));          // byte / 255.0f, bit for bit (vrt_common.hip.h), without a table in LDS
        rec.y = unorm_of((float)((colj >> 8) & 0xffu));
        rec.z = unorm_of((float)((colj >> 16) & 0xffu));
        rec.w = __int_as_float(vidj);
        s_rec[ly * kStride<PX> + slot<PX>(lx)] = rec;
        if (vidj != 0) {   // fold this tap's row and column into its id's ranges, if one of the tile's own pixels carries that id
            if (vidj == run_id) {
                run_lo = ly < run_lo ? ly : run_lo;
                run_hi = ly > run_hi ? ly : run_hi;
                run_xlo = lx < run_xlo ? lx : run_xlo;
                run_xhi = lx > run_xhi ? lx : run_xhi;
            } else {
                flush_run();
                run_id = vidj;
                run_slot = -1;
                run_lo = run_hi = ly;
                run_xlo = run_xhi = lx;
                uint32_t sl = id_slot(vidj);
                for (int tries = 0; tries < kIdSlots; ++tries, sl = (sl + 1u) & (kIdSlots - 1)) {
                    const int at = s_ids->id[sl];
                    if (at == vidj) { run_slot = (int)sl; break; }
                    if (at == 0) break;
                }
            }
        }
    };
    // Four consecutive taps of a row per thread and step where the images allow 16-byte loads (the width a multiple of four, the
    // pointers aligned: tx0 and the 72-tap rows are multiples of four, so a quad lies wholly inside or outside the image): one
    // 16-byte load for four colours, two for four (voxelID, dist) pairs -- 12 loads per thread instead of 32 -- and four taps that mostly
    // carry ONE id, so the id table sees one run per quad instead of one per tap (its atomics were a fifth of the pass without its sums).
    // (Measured and not kept: the quad's four first probes of the table read together and atomics only where a tap widens the bound it
    // reads -- same time: what the staging phase waits for is LDS itself, busy with the other work-group's walk.)
    constexpr int kQuads = kTaps / 4, kQuadIters = (kQuads + kThreads - 1) / kThreads;
    static_assert(kSpanX % 4 == 0 && kTW % 4 == 0 && kMaxR % 4 == 0, "quads never straddle a row or the image's edge");
    const bool quads = (a.width & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.rgba) | reinterpret_cast<uintptr_t>(a.id)) & 15u) == 0u;   // uniform
    if (quads) {
        uint4 cq[kQuadIters], iq0[kQuadIters], iq1[kQuadIters];
#pragma unroll
        for (int q = 0; q < kQuadIters; ++q) {
            const int i4 = tid + q * kThreads;
            const int lx = (i4 % (kSpanX / 4)) * 4, ly = i4 / (kSpanX / 4);
            const int gx = tx0 + lx, gy = ty0 + ly;
            cq[q] = iq0[q] = iq1[q] = make_uint4(0u, 0u, 0u, 0u);
            if (i4 < kQuads && gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
                const size_t g = (size_t)gy * (size_t)a.width + (size_t)gx;
                cq[q] = *reinterpret_cast<const uint4 *>(a.rgba + g);
                iq0[q] = *reinterpret_cast<const uint4 *>(a.id + g);        // (id, dist) of pixels g, g + 1
                iq1[q] = *reinterpret_cast<const uint4 *>(a.id + g + 2);    // ... g + 2, g + 3
            }
        }
#ifdef VRT_DENOISE_PHASE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        VRT_PH(32);   // the window's loads requested and arrived
#pragma unroll
        for (int q = 0; q < kQuadIters; ++q) {
            const int i4 = tid + q * kThreads;
            if (i4 < kQuads) {
                const int lx = (i4 % (kSpanX / 4)) * 4, ly = i4 / (kSpanX / 4);
                stage_tap((int)iq0[q].x, cq[q].x, lx, ly);
                stage_tap((int)iq0[q].z, cq[q].y, lx + 1, ly);
                stage_tap((int)iq1[q].x, cq[q].z, lx + 2, ly);
                stage_tap((int)iq1[q].z, cq[q].w, lx + 3, ly);
            }
        }
    } else
    for (int b = 0; b < kStageIters; b += kBatch) {
        int vid[kBatch];
        uint32_t col[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int i = tid + (b + j) * kThreads;
            const int lx = i % kSpanX, ly = i / kSpanX;
            const int gx = tx0 + lx, gy = ty0 + ly;
            vid[j] = 0;
            col[j] = 0u;
            if (i < kTaps && gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
                const size_t g = (size_t)gy * (size_t)a.width + (size_t)gx;
                vid[j] = a.id[g].x;
                col[j] = a.rgba[g];
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int i = tid + (b + j) * kThreads;
            if (i < kTaps) stage_tap(vid[j], col[j], i % kSpanX, i / kSpanX);
        }
    }
    flush_run();
    VRT_PH(3);   // staging
    __syncthreads();
    VRT_PH(4);   // the barrier behind it

#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int h = __shfl_xor(r_hi, off), l = __shfl_xor(r_lo, off);
        r_hi = h > r_hi ? h : r_hi;
        r_lo = l < r_lo ? l : r_lo;
    }
    r_hi = __builtin_amdgcn_readfirstlane(r_hi);
    r_lo = __builtin_amdgcn_readfirstlane(r_lo);

    Acc<PX> acc;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        acc.rg[k] = f2{0.0f, 0.0f};
        acc.bc[k] = f2{0.0f, 0.0f};
    }
    if (r_hi != 0) {  // some pixel of this wave is summed
        const int rm = r_hi, delta = r_hi - r_lo;
        // the rows this wave has to walk: from the lowest first row to the highest last row of its lanes' ids, inside the window
        int y_first = rm, y_last = -rm;   // relative to the lane's own row (all lanes of a wave row share threadIdx.y: window row y is staged row threadIdx.y + kMaxR + y)
        int u_first = 2 * kMaxR + PX, u_last = -1;   // tap index of the lane's rows: tap u is staged column kMaxR - rm + u + threadIdx.x * PX
        if (s_ids->overflow) { y_first = -rm; y_last = rm; u_first = 0; u_last = 2 * rm + PX - 1; }
        else {
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                if (cid[k] == 0) continue;
                uint32_t sl = id_slot(cid[k]);
                while (s_ids->id[sl] != cid[k]) sl = (sl + 1u) & (kIdSlots - 1);   // present: claimed above
                const int lo = s_ids->lo[sl] - (int)(threadIdx.y + kMaxR), hi = s_ids->hi[sl] - (int)(threadIdx.y + kMaxR);
                const int f = lo < -R[k] ? -R[k] : lo, l = hi > R[k] ? R[k] : hi;
                y_first = f < y_first ? f : y_first;
                y_last = l > y_last ? l : y_last;
                // columns: pixel k sits at tap index rm + k of its lane's rows; its id occurs in staged columns [xlo, xhi]
                const int shift = kMaxR - rm + (int)threadIdx.x * PX;
                const int cf = s_ids->xlo[sl] - shift, cl = s_ids->xhi[sl] - shift;
                const int uf = cf < rm + k - R[k] ? rm + k - R[k] : cf, ul = cl > rm + k + R[k] ? rm + k + R[k] : cl;
                u_first = uf < u_first ? uf : u_first;
                u_last = ul > u_last ? ul : u_last;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int f = __shfl_xor(y_first, off), l = __shfl_xor(y_last, off);
                y_first = f < y_first ? f : y_first;
                y_last = l > y_last ? l : y_last;
                const int uf = __shfl_xor(u_first, off), ul = __shfl_xor(u_last, off);
                u_first = uf < u_first ? uf : u_first;
                u_last = ul > u_last ? ul : u_last;
            }
            y_first = __builtin_amdgcn_readfirstlane(y_first);
            y_last = __builtin_amdgcn_readfirstlane(y_last);
            u_first = __builtin_amdgcn_readfirstlane(u_first);
            u_last = __builtin_amdgcn_readfirstlane(u_last);
        }
        // the segments of kSeg taps that hold a tap some lane needs
        uint32_t seg_mask = 0u;
        for (int sg = 0; sg * kSeg < 2 * rm + PX; ++sg)
            if (sg * kSeg <= u_last && sg * kSeg + kSeg - 1 >= u_first) seg_mask |= 1u << sg;
        VRT_PH(5);   // radii and common ranges across the wave
        // every pixel's own box (rows_own_box) and what the wave's loops over them would cost against the common rows
        bool own = false;
        if (!s_ids->overflow && a.rows_path != 2) {
            Boxes<PX> bx_;
            int hm = 0, wm = 0;   // the largest box among the lane's pixels, then among the wave's lanes
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                bx_.cid[k] = cid[k]; bx_.ry0[k] = bx_.cx0[k] = 0;
                bx_.h[k] = bx_.w[k] = 0;
                if (cid[k] != 0) {
                    uint32_t sl = id_slot(cid[k]);
                    while (s_ids->id[sl] != cid[k]) sl = (sl + 1u) & (kIdSlots - 1);
                    const int row_s = (int)threadIdx.y + kMaxR, col_s = kMaxR + (int)threadIdx.x * PX + k;
                    const int r0 = s_ids->lo[sl] > row_s - R[k] ? s_ids->lo[sl] : row_s - R[k];
                    const int r1 = s_ids->hi[sl] < row_s + R[k] ? s_ids->hi[sl] : row_s + R[k];
                    const int c0 = s_ids->xlo[sl] > col_s - R[k] ? s_ids->xlo[sl] : col_s - R[k];
                    const int c1 = s_ids->xhi[sl] < col_s + R[k] ? s_ids->xhi[sl] : col_s + R[k];
                    bx_.ry0[k] = r0; bx_.h[k] = r1 - r0 + 1;    // the pixel itself carries the id: both at least 1
                    bx_.cx0[k] = c0; bx_.w[k] = c1 - c0 + 1;
                }
                hm = bx_.h[k] > hm ? bx_.h[k] : hm;
                wm = bx_.w[k] > wm ? bx_.w[k] : wm;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const int hh = __shfl_xor(hm, off), ww = __shfl_xor(wm, off);
                hm = hh > hm ? hh : hm;
                wm = ww > wm ? ww : wm;
            }
            hm = __builtin_amdgcn_readfirstlane(hm);
            wm = __builtin_amdgcn_readfirstlane(wm);
            const int cost_own = hm * (((wm + 7) & ~7) * 5 * PX + 8);
            const int cost_common = (y_last - y_first + 1) * (__builtin_popcount(seg_mask) * kSeg * 4 * PX + 4);
            own = a.rows_path == 3 || cost_own * 9 < cost_common * 8;
            if (own) {
                VRT_PH(6);   // boxes, prices
                acc = rows_own_box<PX>((LdsF4)s_rec, bx_, hm, wm);
                VRT_PH(7);   // the own-box walk
            }
        }
        const f4 *row = s_rec + (threadIdx.y + kMaxR - rm) * kStride<PX> + threadIdx.x;  // window row -rm, lane column base
        if (own) {
        } else if (delta == 0)
            rows_dispatch<PX, 0>(row, rm, cid, R, acc, y_first, y_last, seg_mask);
        else if (delta == 1)
            rows_dispatch<PX, 1>(row, rm, cid, R, acc, y_first, y_last, seg_mask);
        else
            rows_dispatch<PX, kFull>(row, rm, cid, R, acc, y_first, y_last, seg_mask);
    }
    VRT_PH(8);   // the common-rows walk (waves that took it)
    if (py >= a.height) return;
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const int px = px0 + k;
        if (px >= a.width) break;
        const size_t p = (size_t)py * (size_t)a.width + (size_t)px;
        if (cid[k] == 0) {  // quad.frag:36-39
            a.out[p] = a.rgba[p];
            continue;
        }
        const float d = fmax_c(acc.bc[k].y, 1.0f);
        a.out[p] = unorm8(acc.rg[k].x / d) | (unorm8(acc.rg[k].y / d) << 8) | (unorm8(acc.bc[k].x / d) << 16) | (255u << 24);
    }
#ifdef VRT_DENOISE_PHASE
    VRT_PH(9);   // divisions and stores
    if (tid == 0) a.out[(size_t)py * (size_t)a.width + (size_t)px0] = ph_ticks;
#endif
}


// One workgroup per tile. waves_per_eu(2, 2) keeps an instance within 256 registers: the default kernel (PX = 2,
// TH = 16: four waves per workgroup, two workgroups per CU by LDS) needs two waves to share a SIMD. Without it the
// compiler, seeing occupancy already limited by LDS, spreads into AGPRs and the second workgroup no longer fits.
template <int PX, int TH, bool SCHED = false>
__global__ __launch_bounds__(kTW / PX *TH) __attribute__((amdgpu_waves_per_eu(2, 2))) void denoise_px_kernel(const Args a) {
    __shared__ f4 s_rec[kStride<PX> * (TH + 2 * kMaxR + 1)];   // + one row of zeros behind the window (rows_own_box reads past its boxes)
    __shared__ IdRows s_ids;
    if constexpr (!SCHED) {
        tile<PX, TH>(a, blockIdx.x, blockIdx.y, s_rec, &s_ids);
    } else {  // 1-D grid of whole groups; the tiles of a frame differ by two orders of magnitude (sky: a copy; radius 20: 1,681 taps)
        int t = (int)blockIdx.x;
        if (a.group_order) t = (int)a.group_order[blockIdx.x / kGroupTiles] * kGroupTiles + (int)(blockIdx.x % kGroupTiles);
        if (t >= a.n_tiles) return;
        const unsigned long long t_begin = a.tile_cost ? __builtin_readcyclecounter() : 0ull;
        const int by = t / a.tiles_x;
        tile<PX, TH>(a, t - by * a.tiles_x, by, s_rec, &s_ids);
        if (a.tile_cost && ((threadIdx.y * (kTW / PX) + threadIdx.x) & 63) == 0)
            atomicMax(&a.tile_cost[t], (uint32_t)(__builtin_readcyclecounter() - t_begin));
    }
}

}  // namespace denoise
}  // namespace vrt
