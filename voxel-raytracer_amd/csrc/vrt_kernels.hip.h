// vrt_kernels.hip.h -- traversal variant "v2" (the default): bit-indexed descent.
//
// The PMC profile of v1 (profiles/r01_a_pmc_summary_v1_variant1.txt) shows the
// path is instruction-issue bound on gfx950 (99.7 % L1 hits, ~4300 VALU
// instructions per wave), so v2 attacks instruction count, not bytes:
//
//  * Power-of-two aligned sub-trees. Below any node whose AABB is a cube of side
//    2^s with its corner on a multiple of 2^s, every descendant is such a cube
//    too, so child selection is three bit extracts of the query point and the
//    node AABB is recovered from the point and the level alone (min = p & ~(2^s-1)).
//    For the reference's world [-1023,1024)^3 the octant [0,1024)^3 that holds
//    every shipped scene is aligned, as are all but the outermost shells of the
//    other octants; the generic explicit-AABB descent (identical to v1) runs only
//    for those top one to three levels.
//  * Restart points. octreeFind restarts at the root when the point leaves the
//    cached parent (23 % of finds, 11 levels each on dragon.vox). Here each ray
//    also keeps an "anchor": its ancestor of side >= 2^kAnchorShift. Leaving the
//    parent but not the anchor (the common case) costs <= kAnchorShift levels.
//    The lookup result is a pure function of the point, so where the descent
//    starts cannot change any output.
//  * The refraction index of a leaf (byte/255*3) comes from a 256-entry table
//    built in LDS by each workgroup with the same correctly rounded operations,
//    instead of a division per DDA step.
//  * Optionally the level-order prefix of the record array (the hot top of the
//    tree) is staged in LDS as well (USE_LDS).
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace v2 {

constexpr int kAnchorShift = 5;

struct Walk {                  // per-ray lookup state carried from one find to the next
    uint32_t pm, pb; int ps;   // cached parent: masks, first-child index, log2(side); ps < 0: none
    uint32_t am, ab; int as;   // anchor ancestor, same encoding
    I3 last;                   // the previous query point (inside parent and anchor)
};

struct Found {
    uint32_t w0, w1;           // leaf words, or 0/0 for empty space
    I3 mn, mx;                 // AABB of the node found
};

template <bool USE_LDS>
struct Trav {
    struct Ctx {
        const uint2 *lds;      // staged record prefix (USE_LDS)
        const float *refr;     // 256-entry refraction table in LDS
        uint2 root;
    };

    template <int BLOCK>
    static VRT_DEV void block_init(const KArgs &a, uint2 *lds_dyn, Ctx &c) {
        __shared__ float refr_lut[256];
        for (int i = threadIdx.x; i < 256; i += BLOCK) refr_lut[i] = ((float)i / 255.0f) * 3.0f;  // comp:126-128,177
        if (USE_LDS) {
            for (uint32_t i = threadIdx.x; i < a.lds_records; i += BLOCK) lds_dyn[i] = a.nodes[i];
        }
        __syncthreads();
        c.lds = lds_dyn;
        c.refr = refr_lut;
        c.root = a.nodes[0];
    }

    static VRT_DEV uint2 load_record(const KArgs &a, const Ctx &c, uint32_t idx) {
        if (USE_LDS) {
            if (idx < a.lds_records) return c.lds[idx];
        }
        return a.nodes[idx];
    }

    static VRT_DEV void reset(Walk &w) { w.ps = -1; w.as = -1; w.pm = w.pb = w.am = w.ab = 0u; w.last = I3{0, 0, 0}; }

    // octreeFind (comp:137-220) for a point known to be inside the world.
    static VRT_DEV Found find_node(const KArgs &a, const Ctx &c, I3 p, Walk &w) {
        Found f;
        f.w0 = 0u; f.w1 = 0u;
        uint32_t m = 0u, b = 0u;
        int s = -1;
        if (w.ps >= 0) {
            const uint32_t d = (uint32_t)((p.x ^ w.last.x) | (p.y ^ w.last.y) | (p.z ^ w.last.z));
            if ((d >> w.ps) == 0u) { m = w.pm; b = w.pb; s = w.ps; }
            else if ((d >> w.as) == 0u) { m = w.am; b = w.ab; s = w.as; }   // as >= ps >= 0 whenever ps >= 0
        }
        if (s < 0) {
            // generic descent from the root with explicit AABBs until an aligned cube is reached
            m = c.root.x; b = c.root.y;
            I3 mn{a.wmin[0], a.wmin[1], a.wmin[2]}, mx{a.wmax[0], a.wmax[1], a.wmax[2]};
            w.ps = -1; w.as = -1;
            for (int i = 0; i < 16; ++i) {
                const int sx = mx.x - mn.x;
                if (sx == mx.y - mn.y && sx == mx.z - mn.z && sx > 0 && sx <= (1 << 30) && (sx & (sx - 1)) == 0 &&
                    (((mn.x | mn.y | mn.z) & (sx - 1)) == 0)) {
                    s = 31 - __builtin_clz((unsigned)sx);
                    w.am = m; w.ab = b; w.as = s;
                    break;
                }
                const int cx = mn.x + ((mx.x - mn.x) >> 1), cy = mn.y + ((mx.y - mn.y) >> 1), cz = mn.z + ((mx.z - mn.z) >> 1);
                const bool hx = p.x >= cx, hy = p.y >= cy, hz = p.z >= cz;
                const uint32_t ci = (hx ? 4u : 0u) | (hy ? 2u : 0u) | (hz ? 1u : 0u);
                mn = I3{hx ? cx : mn.x, hy ? cy : mn.y, hz ? cz : mn.z};
                mx = I3{hx ? mx.x : cx, hy ? mx.y : cy, hz ? mx.z : cz};
                const uint32_t bit = 1u << ci;
                if (!(m & bit)) { f.mn = mn; f.mx = mx; return f; }
                const uint2 rec = load_record(a, c, b + (uint32_t)__builtin_popcount(m & 0xffu & (bit - 1u)));
                if (m & (bit << 8)) { f.w0 = rec.x; f.w1 = rec.y; f.mn = mn; f.mx = mx; return f; }
                m = rec.x; b = rec.y;
            }
            if (s < 0) { f.mn = mn; f.mx = mx; return f; }  // deeper than the uploader allows: treated as empty
        }
        // bit-indexed descent inside an aligned cube of side 2^s
        int s1;
        for (;;) {
            uint32_t ci;
            if (s == 0) { s1 = 0; ci = 7u; }  // a unit cell that is still internal (never written by octree_texture)
            else {
                s1 = s - 1;
                ci = ((((uint32_t)p.x >> s1) & 1u) << 2) | ((((uint32_t)p.y >> s1) & 1u) << 1) | (((uint32_t)p.z >> s1) & 1u);
            }
            const uint32_t bit = 1u << ci;
            if (!(m & bit)) break;
            const uint2 rec = load_record(a, c, b + (uint32_t)__builtin_popcount(m & 0xffu & (bit - 1u)));
            if (m & (bit << 8)) { f.w0 = rec.x; f.w1 = rec.y; break; }
            m = rec.x; b = rec.y; s = s1;
            if (s >= kAnchorShift) { w.am = m; w.ab = b; w.as = s; }
        }
        w.pm = m; w.pb = b; w.ps = s; w.last = p;
        const int side = 1 << s1, keep = ~(side - 1);
        f.mn = I3{p.x & keep, p.y & keep, p.z & keep};
        f.mx = I3{f.mn.x + side, f.mn.y + side, f.mn.z + side};
        return f;
    }

    // comp:143-145: outside the world octreeFind returns zeroed data (AABB: convention C8 = world bounds)
    static VRT_DEV Found find_checked(const KArgs &a, const Ctx &c, I3 p, Walk &w) {
        if (!in_world(a, p)) {
            Found f;
            f.w0 = 0u; f.w1 = 0u;
            f.mn = I3{a.wmin[0], a.wmin[1], a.wmin[2]};
            f.mx = I3{a.wmax[0], a.wmax[1], a.wmax[2]};
            return f;
        }
        return find_node(a, c, p, w);
    }

    static VRT_DEV void eye_medium(const KArgs &a, const Ctx &c, I3 p, uint32_t &w0, uint32_t &w1) {
        Walk w;
        reset(w);
        Found f = find_checked(a, c, p, w);
        w0 = f.w0; w1 = f.w1;
    }

    // hitMarching (comp:248-330)
    static VRT_DEV bool march(const KArgs &a, const Ctx &c, F3 origin, F3 dir, float ray_iof, Hit &h) {
        F3 rp = origin;
        float inv_len = 1.0f / __builtin_sqrtf(dot3(dir, dir));
        dir = scale3(dir, inv_len);
        F3 inv;
        inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
        inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
        inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
        const bool px = dir.x > 0.0f, py = dir.y > 0.0f, pz = dir.z > 0.0f;
        // -sign(dir) per axis and the signed 1e-4 push (comp:294,300-304)
        const F3 sd{sign_c(dir.x), sign_c(dir.y), sign_c(dir.z)};
        const F3 push{sd.x * 0.0001f, sd.y * 0.0001f, sd.z * 0.0001f};
        Walk w;
        reset(w);
        I3 mp = floor_i3(rp);
        Found cur = find_checked(a, c, mp, w);
        uint32_t rb = cur.w1 & 0xffu;
        float cur_ref = rb ? c.refr[rb] : 0.0f;
        bool cur_solid = (cur.w0 >> 24) != 0u && cur_ref > 0.0f;
        int axis = 0;
        bool hit = false;
        uint32_t pw0 = 0u, pw1 = 0u;
        for (int i = 0; i < 1024; ++i) {
            const float tx = ((px ? (float)cur.mx.x : (float)cur.mn.x) - rp.x) * inv.x;
            const float ty = ((py ? (float)cur.mx.y : (float)cur.mn.y) - rp.y) * inv.y;
            const float tz = ((pz ? (float)cur.mx.z : (float)cur.mn.z) - rp.z) * inv.z;
            const float t = fmin_c(tx, fmin_c(ty, tz));
            axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
            rp.x = rp.x + dir.x * t; rp.y = rp.y + dir.y * t; rp.z = rp.z + dir.z * t;
            if (axis == 0) rp.x = rp.x + push.x; else if (axis == 1) rp.y = rp.y + push.y; else rp.z = rp.z + push.z;
            mp = floor_i3(rp);
            if (!in_world(a, mp)) break;
            pw0 = cur.w0; pw1 = cur.w1;
            const float prev_ref = cur_solid ? cur_ref : ray_iof;
            cur = find_node(a, c, mp, w);
            rb = cur.w1 & 0xffu;
            cur_ref = rb ? c.refr[rb] : 0.0f;
            cur_solid = (cur.w0 >> 24) != 0u && cur_ref > 0.0f;
            const float now_ref = cur_solid ? cur_ref : 1.0f;
            if (__builtin_fabsf(now_ref - prev_ref) > 0.0001f) { hit = true; break; }
        }
        const float n = -comp(sd, axis);
        h.normal = F3{axis == 0 ? n : 0.0f, axis == 1 ? n : 0.0f, axis == 2 ? n : 0.0f};
        h.map = mp; h.point = rp; h.p0 = pw0; h.p1 = pw1; h.h0 = cur.w0; h.h1 = cur.w1;
        return hit;
    }

    // notInShadow (comp:333-377); the light direction is used as given
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, F3 ld) {
        F3 rp = origin, inv;
        inv.x = (__builtin_fabsf(ld.x) < 1e-8f) ? 1e20f : 1.0f / ld.x;
        inv.y = (__builtin_fabsf(ld.y) < 1e-8f) ? 1e20f : 1.0f / ld.y;
        inv.z = (__builtin_fabsf(ld.z) < 1e-8f) ? 1e20f : 1.0f / ld.z;
        const bool px = ld.x > 0.0f, py = ld.y > 0.0f, pz = ld.z > 0.0f;
        const F3 push{sign_c(ld.x) * 0.001f, sign_c(ld.y) * 0.001f, sign_c(ld.z) * 0.001f};
        I3 mp = floor_i3(rp);
        Walk w;
        reset(w);
        Found v = find_checked(a, c, mp, w);
        for (int i = 0; i < 64; ++i) {
            // occluder: alpha > 0.1 (alpha byte / 255) and illumination byte == 0 (comp:355)
            const float alpha = (float)(v.w0 >> 24) / 255.0f;
            if (alpha > 0.1f && ((v.w1 >> 8) & 0xffu) == 0u) return 0;
            const float tx = ((px ? (float)v.mx.x : (float)v.mn.x) - rp.x) * inv.x;
            const float ty = ((py ? (float)v.mx.y : (float)v.mn.y) - rp.y) * inv.y;
            const float tz = ((pz ? (float)v.mx.z : (float)v.mn.z) - rp.z) * inv.z;
            const float t = fmin_c(tx, fmin_c(ty, tz));
            const int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
            rp.x = rp.x + ld.x * t; rp.y = rp.y + ld.y * t; rp.z = rp.z + ld.z * t;
            if (axis == 0) rp.x = rp.x + push.x; else if (axis == 1) rp.y = rp.y + push.y; else rp.z = rp.z + push.z;
            mp = floor_i3(rp);
            if (!in_world(a, mp)) return 1;
            if (i == 63) break;  // the 64th iteration's find result is never inspected
            v = find_node(a, c, mp, w);
        }
        return 1;
    }
};

}  // namespace v2
}  // namespace vrt
