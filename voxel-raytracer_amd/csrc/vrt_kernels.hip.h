// vrt_kernels.hip.h -- traversal variant "v2": bit-indexed descent over the record array. The fallback
// when a scene has no wide form (vrt_kernels_wide.hip.h, the default), and the middle rung of the A/B ladder.
//
// The PMC profile of v1 (profiles/r01_a_pmc_summary_v1_variant1.txt) shows the
// path is instruction-issue bound on gfx950 (99.7 % L1 hits, ~4300 VALU
// instructions per wave), so v2 attacks instruction count, not bytes:
//
//  * Power-of-two aligned sub-trees. Below any node whose AABB is a cube of side
//    2^s with its corner on a multiple of 2^s, every descendant is such a cube
//    too, so child selection is three bit extracts of the query point and the
//    node AABB is recovered from the point and the level alone (min = p & -2^s).
//    For the reference's world [-1023,1024)^3 the octant [0,1024)^3 that holds
//    every shipped scene is aligned, as are all but the outermost shells of the
//    other octants; the generic explicit-AABB descent (identical to v1) runs only
//    for those top one to three levels.
//  * Restart points. octreeFind restarts at the root when the point leaves the
//    cached parent (a quarter of the finds, 11 levels each on dragon.vox). Here
//    each ray also keeps an "anchor": its ancestor of side >= 2^kAnchorShift.
//    Leaving the parent but not the anchor (the common case) costs at most
//    kAnchorShift levels. The lookup result is a pure function of the point, so
//    where the descent starts cannot change any output.
//  * Single-exit, predicated loops. Both the descent and the DDA loop are written
//    as do-while loops with one continuation flag and selects for the state
//    updates; the multi-exit form costs ~20 scalar mask instructions per
//    iteration on gfx950 (measured in the v2a ISA).
//  * The medium-change test of hitMarching (comp:318-321) is evaluated on the
//    refraction BYTES instead of on floats; see medium_byte() for the proof of
//    equivalence. No division and no table per DDA step.
//  * Optionally the level-order prefix of the record array (the hot top of the
//    tree) is staged in LDS (USE_LDS).
//
// Precondition checked by the host before choosing this variant: no internal
// node is an aligned cube of side 1 (octree_texture never writes one); scenes
// that violate it are traced by the v1 kernels.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace v2 {

constexpr int kAnchorShift = 5;

struct Walk {                  // per-ray lookup state carried from one find to the next
    uint32_t pm, pb; int ps;   // cached parent: masks, first-child index, log2(side); ps < 0: none
    uint32_t am, ab; int as;   // anchor ancestor, same encoding (valid whenever ps >= 0)
    I3 last;                   // the previous query point (inside parent and anchor)
};

struct Found {
    uint32_t w0, w1;           // leaf words, or 0/0 for empty space
    I3 mn, mx;                 // AABB of the node found
};

// Refraction byte of the medium a node represents, 0 when the node does not count as a medium
// (comp:318-319: color.a > 0 && properties[0] > 0  <=>  alpha byte != 0 && refraction byte != 0).
VRT_DEV uint32_t medium_byte(uint32_t w0, uint32_t w1) { return (w0 >> 24) != 0u ? (w1 & 0xffu) : 0u; }
// hitMarching compares refraction indices r(b) = (b/255)*3 as floats: |r_now - r_prev| > 1e-4, with
// 1.0 standing in for "no medium" on the new side and rayIOF on the old side. r is strictly
// increasing in b with steps of 0.01176 > 1e-4, r(85) == 1.0f exactly, and rayIOF is r(b) for some
// b in 1..254 or 1.0 (comp:448-449). Hence the float test is true exactly when the two bytes
// differ, once "no medium" is replaced by 85 (new side) or by the byte of rayIOF (old side).

template <bool USE_LDS>
struct Trav {
    static constexpr bool kStagesLds = USE_LDS;
    struct Ctx {
        const uint2 *lds;      // staged record prefix (USE_LDS)
        uint2 root;
    };

    template <int BLOCK>
    static VRT_DEV void block_init(const KArgs &a, uint2 *lds_dyn, Ctx &c) {
        if (USE_LDS) {
            for (uint32_t i = threadIdx.x; i < a.lds_records; i += BLOCK) lds_dyn[i] = a.nodes[i];
            __syncthreads();
        }
        c.lds = lds_dyn;
        c.root = a.nodes[0];
    }

    static VRT_DEV uint2 load_record(const KArgs &a, const Ctx &c, uint32_t idx) {
        if (USE_LDS) {
            if (idx < a.lds_records) return c.lds[idx];
        }
        return a.nodes[idx];
    }

    static VRT_DEV void reset(Walk &w) { w.ps = -1; w.as = -1; w.pm = w.pb = w.am = w.ab = 0u; w.last = I3{0, 0, 0}; }

    // Generic explicit-AABB descent from the root (the shader's own arithmetic, comp:161-216) until the
    // node reached is an aligned cube. Returns true when the lookup finished here (result in f).
    static VRT_DEV bool descend_generic(const KArgs &a, const Ctx &c, I3 p, Walk &w, Found &f, uint32_t &m, uint32_t &b, int &s) {
        m = c.root.x; b = c.root.y;
        I3 mn{a.wmin[0], a.wmin[1], a.wmin[2]}, mx{a.wmax[0], a.wmax[1], a.wmax[2]};
        w.ps = -1; w.as = -1;
        for (int i = 0; i < 16; ++i) {
            const int sx = mx.x - mn.x;
            if (sx == mx.y - mn.y && sx == mx.z - mn.z && sx > 1 && sx <= (1 << 30) && (sx & (sx - 1)) == 0 &&
                (((mn.x | mn.y | mn.z) & (sx - 1)) == 0)) {
                s = 31 - __builtin_clz((unsigned)sx);
                w.am = m; w.ab = b; w.as = s;
                return false;
            }
            const int cx = mn.x + ((mx.x - mn.x) >> 1), cy = mn.y + ((mx.y - mn.y) >> 1), cz = mn.z + ((mx.z - mn.z) >> 1);
            const bool hx = p.x >= cx, hy = p.y >= cy, hz = p.z >= cz;
            const uint32_t ci = (hx ? 4u : 0u) | (hy ? 2u : 0u) | (hz ? 1u : 0u);
            mn = I3{hx ? cx : mn.x, hy ? cy : mn.y, hz ? cz : mn.z};
            mx = I3{hx ? mx.x : cx, hy ? mx.y : cy, hz ? mx.z : cz};
            const uint32_t bit = 1u << ci;
            f.mn = mn; f.mx = mx;
            if (!(m & bit)) return true;
            const uint2 rec = load_record(a, c, b + (uint32_t)__builtin_popcount(m & (bit - 1u)));
            if (m & (bit << 8)) { f.w0 = rec.x; f.w1 = rec.y; return true; }
            m = rec.x; b = rec.y;
        }
        return true;  // deeper than the uploader allows: treated as empty
    }

    // octreeFind (comp:137-220) for a point known to be inside the world.
    static VRT_DEV Found find_node(const KArgs &a, const Ctx &c, I3 p, Walk &w) {
        Found f;
        f.w0 = 0u; f.w1 = 0u;
        // where to start: cached parent, else anchor, else root
        const uint32_t d = (uint32_t)((p.x ^ w.last.x) | (p.y ^ w.last.y) | (p.z ^ w.last.z));
        const bool have = w.ps >= 0;
        const bool in_parent = have && (d >> (w.ps & 31)) == 0u;
        const bool in_anchor = have && (d >> (w.as & 31)) == 0u;
        uint32_t m = in_parent ? w.pm : w.am;
        uint32_t b = in_parent ? w.pb : w.ab;
        int s = in_parent ? w.ps : w.as;
        if (!(in_parent || in_anchor)) {
            if (descend_generic(a, c, p, w, f, m, b, s)) return f;
        }
        // bit-indexed descent inside an aligned cube of side 2^s, s >= 1
        int s1;
        uint32_t sh;
        uint2 rec;
        bool go;
        do {
            s1 = s - 1;
            const uint32_t ci = ((((uint32_t)p.x >> s1) & 1u) << 2) | ((((uint32_t)p.y >> s1) & 1u) << 1) | (((uint32_t)p.z >> s1) & 1u);
            sh = m >> ci;  // bit 0: child present, bit 8: child is a leaf
            rec = make_uint2(0u, 0u);
            if (sh & 1u) rec = load_record(a, c, b + (uint32_t)__builtin_popcount(m & ((1u << ci) - 1u)));
            go = (sh & 0x101u) == 1u;
            m = go ? rec.x : m;
            b = go ? rec.y : b;
            s = go ? s1 : s;
            const bool up = go && s1 >= kAnchorShift;
            w.am = up ? rec.x : w.am;
            w.ab = up ? rec.y : w.ab;
            w.as = up ? s1 : w.as;
        } while (go);
        const bool leaf = (sh & 0x101u) == 0x101u;
        f.w0 = leaf ? rec.x : 0u;
        f.w1 = leaf ? rec.y : 0u;
        w.pm = m; w.pb = b; w.ps = s; w.last = p;
        const int side = 1 << s1, keep = -side;
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

    // One DDA step shared by march() and shadow(): leave the node [mn,mx) through the nearest far plane.
    static VRT_DEV int dda_step(F3 &rp, F3 dir, F3 inv, F3 push, bool px, bool py, bool pz, const Found &n) {
        const float tx = ((px ? (float)n.mx.x : (float)n.mn.x) - rp.x) * inv.x;
        const float ty = ((py ? (float)n.mx.y : (float)n.mn.y) - rp.y) * inv.y;
        const float tz = ((pz ? (float)n.mx.z : (float)n.mn.z) - rp.z) * inv.z;
        const float t = fmin_c(tx, fmin_c(ty, tz));
        const int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
        rp.x = rp.x + dir.x * t; rp.y = rp.y + dir.y * t; rp.z = rp.z + dir.z * t;
        const float qx = rp.x + push.x, qy = rp.y + push.y, qz = rp.z + push.z;
        rp.x = axis == 0 ? qx : rp.x;
        rp.y = axis == 1 ? qy : rp.y;
        rp.z = axis == 2 ? qz : rp.z;
        return axis;
    }

    // hitMarching (comp:248-330)
    static VRT_DEV bool march(const KArgs &a, const Ctx &c, F3 origin, F3 dir, float ray_iof, uint32_t iof_byte, Hit &h,
                              const View * = nullptr) {
        (void)ray_iof;
        F3 rp = origin;
        float inv_len = 1.0f / __builtin_sqrtf(dot3(dir, dir));
        dir = scale3(dir, inv_len);
        F3 inv;
        inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
        inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
        inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
        const bool px = dir.x > 0.0f, py = dir.y > 0.0f, pz = dir.z > 0.0f;
        // sign(dir) per axis and the signed 1e-4 push (comp:294,300-304)
        const F3 sd{sign_c(dir.x), sign_c(dir.y), sign_c(dir.z)};
        const F3 push{sd.x * 0.0001f, sd.y * 0.0001f, sd.z * 0.0001f};
        Walk w;
        reset(w);
        I3 mp = floor_i3(rp);
        Found cur = find_checked(a, c, mp, w);
        uint32_t cur_b = medium_byte(cur.w0, cur.w1);
        int axis = 0;
        bool hit = false, go;
        uint32_t pw0 = 0u, pw1 = 0u;
        int i = 0;
        do {
            axis = dda_step(rp, dir, inv, push, px, py, pz, cur);
            mp = floor_i3(rp);
            const bool inw = in_world(a, mp);
            if (inw) {
                pw0 = cur.w0; pw1 = cur.w1;
                const uint32_t prev_b = cur_b ? cur_b : iof_byte;
                cur = find_node(a, c, mp, w);
                cur_b = medium_byte(cur.w0, cur.w1);
                hit = (cur_b ? cur_b : 85u) != prev_b;
            }
            ++i;
            go = inw && !hit && i < 1024;
        } while (go);
        const float n = -comp(sd, axis);
        h.axis = axis; h.n = n;
        h.map = mp; h.point = rp; h.p0 = pw0; h.p1 = pw1; h.h0 = cur.w0; h.h1 = cur.w1;
        return hit;
    }

    // notInShadow (comp:333-377); the light direction is used as given
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, F3 ld, const Hit & /*resume hint unused*/) {
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
        int lit = 1, i = 0;
        bool go;
        do {
            // occluder: alpha > 0.1 <=> alpha byte >= 26 (25/255 = 0.098, 26/255 = 0.102); illumination byte == 0 (comp:355)
            const bool occluder = (v.w0 >> 24) >= 26u && ((v.w1 >> 8) & 0xffu) == 0u;
            lit = occluder ? 0 : lit;
            (void)dda_step(rp, ld, inv, push, px, py, pz, v);
            mp = floor_i3(rp);
            ++i;
            go = !occluder && in_world(a, mp) && i < 64;
            if (go) v = find_node(a, c, mp, w);
        } while (go);
        return lit;
    }
};

}  // namespace v2
}  // namespace vrt
