// vrt_kernels_wide.hip.h -- traversal variant "v3" (the default): dense 64-cell wide nodes.
//
// PMC of v1/v2 (profiles/): the path is bound by VALU issue on gfx950 (a wave64
// VALU instruction occupies its SIMD for 4 cycles; 99.7 % of node reads hit L1),
// and most of those instructions are spent walking octree levels: 2.5 levels per
// lookup for a lane, ~3.8 per lookup for a wave because every DDA step waits for
// its slowest lane. v3 replaces the level walk inside aligned sub-trees by the
// wide layout of vrt_layout.h: one direct-indexed 8-byte cell load per TWO octree
// levels, and the cell already carries the lookup's answer (leaf words + the
// log2 size of the octree node the point is in). A lookup that stays inside the
// ray's current wide node -- the common case, a wide node spans 4x4x4 cells -- is
// one load and ~20 VALU instructions with no loop-carried divergence.
//
// The result of a lookup is still exactly octreeFind's (comp:137-220): the deepest
// octree node containing the point, its leaf data and its AABB; the DDA
// (comp:277-327) is untouched. Where the descent starts cannot change any output.
//
// Uses the unit-internal-node precondition of vrt_kernels.hip.h and at most
// 8 wide roots; the host falls back to v2/v1 otherwise.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace v3 {

// kAnchorShift (vrt_args.h): the restart point, the wide node of side 64 the ray is in

struct Walk {                 // per-ray lookup state carried from one find to the next
    uint32_t node; int s;     // current wide node and log2 of its side; s < 0: none
    uint32_t anode; int as;   // anchor wide node (an ancestor of `node`, or `node` itself)
    I3 last;                  // the previous query point (inside both)
};

struct Found {
    uint32_t w0, w1;          // leaf words, or 0/0 for empty space
    I3 plane;                 // per axis: the face of the node found that a ray with signs `dpos` leaves through
};

// Medium test on refraction bytes: see medium_byte() in vrt_kernels.hip.h. Here the byte is prepared at
// build time: every leaf word pair a lookup returns has refraction byte 0 when its alpha byte is 0
// (wide cells: vrt_layout.cpp; record leaves: descend_generic). Nothing downstream can tell: the shader
// replaces the properties of an alpha-0 voxel before using them (comp:503-504); only the lookup at the
// eye reads them unconditionally, and that one is made by the dispatcher on the raw records (eye_lookup()).

// (int)floor(x) in one instruction
VRT_DEV int floor_to_int(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
VRT_DEV I3 floor_i3_fast(F3 p) { return I3{floor_to_int(p.x), floor_to_int(p.y), floor_to_int(p.z)}; }

VRT_DEV bool in_world_u(const KArgs &a, I3 p) {  // comp:224-226, as three unsigned range checks
    return (uint32_t)(p.x - a.wmin[0]) < (uint32_t)(a.wmax[0] - a.wmin[0]) &&
           (uint32_t)(p.y - a.wmin[1]) < (uint32_t)(a.wmax[1] - a.wmin[1]) &&
           (uint32_t)(p.z - a.wmin[2]) < (uint32_t)(a.wmax[2] - a.wmin[2]);
}

struct Trav {
    static constexpr bool kStagesLds = false;
    struct Ctx {
        uint2 root;
    };

    template <int BLOCK>
    static VRT_DEV void block_init(const KArgs &a, uint2 *, Ctx &c) { c.root = a.nodes[0]; }

    static VRT_DEV void reset(Walk &w) { w.s = -1; w.as = -1; w.node = w.anode = 0u; w.last = I3{0, 0, 0}; }

    // Explicit-AABB descent from the octree root over the record array (the shader's arithmetic,
    // comp:161-216) until a wide root is reached. Returns true when the lookup finished here.
    static VRT_DEV bool descend_generic(const KArgs &a, const Ctx &c, I3 p, I3 dpos, Walk &w, Found &f, uint32_t &node, int &s) {
        uint32_t m = c.root.x, b = c.root.y, ridx = 0u;
        I3 mn{a.wmin[0], a.wmin[1], a.wmin[2]}, mx{a.wmax[0], a.wmax[1], a.wmax[2]};
        w.s = -1; w.as = -1;
        for (int i = 0; i < 16; ++i) {
            const int sx = mx.x - mn.x;
            if (sx == mx.y - mn.y && sx == mx.z - mn.z && sx >= 4 && sx <= (1 << 30) && (sx & (sx - 1)) == 0 &&
                (((mn.x | mn.y | mn.z) & (sx - 1)) == 0)) {
                const int sh = 31 - __builtin_clz((unsigned)sx);
                if (!(sh & 1)) {
                    bool found = false;
                    // the table is read here, on the rare path, and nowhere else: the empty asm keeps the compiler from
                    // hoisting sixteen scalar loads (and their registers) out of the traversal loop
                    const uint32_t *roots = a.root_table;
                    asm volatile("" : "+s"(roots));
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (roots[k] == ridx) { node = roots[8 + k]; found = true; }  // unused slots hold record 0xffffffff
                    if (found) {
                        s = sh;
                        w.anode = node; w.as = sh;
                        return false;
                    }
                }
            }
            const int cx = mn.x + ((mx.x - mn.x) >> 1), cy = mn.y + ((mx.y - mn.y) >> 1), cz = mn.z + ((mx.z - mn.z) >> 1);
            const bool hx = p.x >= cx, hy = p.y >= cy, hz = p.z >= cz;
            const uint32_t ci = (hx ? 4u : 0u) | (hy ? 2u : 0u) | (hz ? 1u : 0u);
            mn = I3{hx ? cx : mn.x, hy ? cy : mn.y, hz ? cz : mn.z};
            mx = I3{hx ? mx.x : cx, hy ? mx.y : cy, hz ? mx.z : cz};
            f.plane = I3{dpos.x ? mx.x : mn.x, dpos.y ? mx.y : mn.y, dpos.z ? mx.z : mn.z};
            const uint32_t bit = 1u << ci;
            if (!(m & bit)) return true;
            ridx = b + (uint32_t)__builtin_popcount(m & (bit - 1u));
            const uint2 rec = a.nodes[ridx];
            if (m & (bit << 8)) {  // leaf: like the wide cells, report refraction byte 0 when alpha == 0
                f.w0 = rec.x;
                f.w1 = (rec.x >> 24) != 0u ? rec.y : (rec.y & 0xffffff00u);
                return true;
            }
            m = rec.x; b = rec.y;
        }
        return true;  // deeper than the uploader allows: treated as empty
    }

    // octreeFind (comp:137-220) for a point known to be inside the world.
    static VRT_DEV Found find_node(const KArgs &a, const Ctx &c, I3 p, I3 dpos, Walk &w) {
        Found f;
        f.w0 = 0u; f.w1 = 0u;
        const uint32_t d = (uint32_t)((p.x ^ w.last.x) | (p.y ^ w.last.y) | (p.z ^ w.last.z));
        const bool have = w.s >= 0;
        const bool in_node = have && (d >> (w.s & 31)) == 0u;
        const bool in_anchor = have && (d >> (w.as & 31)) == 0u;
        uint32_t node = in_node ? w.node : w.anode;
        int s = in_node ? w.s : w.as;
        if (!(in_node || in_anchor)) {
            // Left the anchor. Inside the cube of wide root 0 (the octant that holds every shipped scene) the
            // descent from the octree root can only end at that root: start there directly, in the state
            // descend_generic() would leave behind. Anywhere else walk the records.
            const int rs = a.root0_shift;
            const uint32_t out0 = (uint32_t)((p.x ^ a.root0_min[0]) | (p.y ^ a.root0_min[1]) | (p.z ^ a.root0_min[2])) >> (rs & 31);
            if (a.n_roots != 0u && out0 == 0u) {
                node = a.root0_node; s = rs;
                w.s = -1; w.anode = node; w.as = rs;
            } else if (descend_generic(a, c, p, dpos, w, f, node, s)) {
                return f;
            }
        }
        uint2 cell;
        bool go;
        do {
            const int cs = s - 2;
            const uint32_t ci = (__builtin_amdgcn_ubfe((uint32_t)p.x, (uint32_t)cs, 2u) << 4) |
                                (__builtin_amdgcn_ubfe((uint32_t)p.y, (uint32_t)cs, 2u) << 2) |
                                __builtin_amdgcn_ubfe((uint32_t)p.z, (uint32_t)cs, 2u);
            cell = a.cells[(node << 6) | ci];
            go = (int)cell.y < 0;  // bit 31: subdivided further, cell.x = child wide node
            node = go ? cell.x : node;
            s = go ? cs : s;
            const bool up = go && cs == kAnchorShift;
            w.anode = up ? cell.x : w.anode;
            w.as = up ? cs : w.as;
        } while (go);
        w.node = node; w.s = s; w.last = p;
        const int t = (int)(cell.y >> 24);  // log2 side of the octree node found (bits 29..31 are clear here)
        f.w0 = cell.x;
        f.w1 = cell.y & 0x00ffffffu;
        f.plane = I3{((p.x >> t) + dpos.x) << t, ((p.y >> t) + dpos.y) << t, ((p.z >> t) + dpos.z) << t};
        return f;
    }

    // comp:143-145: outside the world octreeFind returns zeroed data (AABB: convention C8 = world bounds)
    static VRT_DEV Found find_checked(const KArgs &a, const Ctx &c, I3 p, I3 dpos, Walk &w) {
        if (!in_world_u(a, p)) {
            Found f;
            f.w0 = 0u; f.w1 = 0u;
            f.plane = I3{dpos.x ? a.wmax[0] : a.wmin[0], dpos.y ? a.wmax[1] : a.wmin[1], dpos.z ? a.wmax[2] : a.wmin[2]};
            return f;
        }
        return find_node(a, c, p, dpos, w);
    }

    // One DDA step (comp:278-307): leave the current node through its nearest far plane.
    // The exit axis (comp:292) is kept as two predicates: x = tx<ty && tx<tz, y = !(tx<ty) && ty<tz, else z.
    // tStep = min(tMax.x, min(tMax.y, tMax.z)) (comp:291) equals the tMax of that axis: where two of them tie
    // the values are equal, so picking either is the same number.
    struct Axis { bool x, y; };
    static VRT_DEV Axis dda_step(F3 &rp, F3 dir, F3 inv, F3 push, I3 plane) {
        typedef float f2v __attribute__((ext_vector_type(2)));
        // x and y travel as one packed pair (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations, ~5 cycles per
        // pair instead of 8), z alone
        const f2v pl = {(float)plane.x, (float)plane.y}, r = {rp.x, rp.y}, iv = {inv.x, inv.y}, d = {dir.x, dir.y},
                  pu = {push.x, push.y};
        const f2v t2 = (pl - r) * iv;
        const float tx = t2.x, ty = t2.y;
        const float tz = ((float)plane.z - rp.z) * inv.z;
        const bool xy = tx < ty, xz = tx < tz, yz = ty < tz;
        const bool ax = xy && xz;
        const bool ay = !xy && yz;
        const bool az = !(ax || ay);
        const float t = ax ? tx : (ay ? ty : tz);
        const f2v tt = {t, t};
        const f2v r2 = r + d * tt;
        const float rz = rp.z + dir.z * t;
        const f2v q2 = r2 + pu;
        const float qz = rz + push.z;
        rp.x = ax ? q2.x : r2.x;
        rp.y = ay ? q2.y : r2.y;
        rp.z = az ? qz : rz;
        return Axis{ax, ay};
    }

    // hitMarching (comp:248-330)
    // eye: the view whose eye `origin` is (primary rays): its first lookup was made by the host
    static VRT_DEV bool march(const KArgs &a, const Ctx &c, F3 origin, F3 dir, float ray_iof, uint32_t iof_byte, Hit &h,
                              const View *eye = nullptr) {
        (void)ray_iof;
        F3 rp = origin;
        float inv_len = 1.0f / __builtin_sqrtf(dot3(dir, dir));
        dir = scale3(dir, inv_len);
        F3 inv;
        inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
        inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
        inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
        const I3 dpos{dir.x > 0.0f ? 1 : 0, dir.y > 0.0f ? 1 : 0, dir.z > 0.0f ? 1 : 0};
        const F3 sd{sign_c(dir.x), sign_c(dir.y), sign_c(dir.z)};
        const F3 push{sd.x * 0.0001f, sd.y * 0.0001f, sd.z * 0.0001f};  // comp:300-304
        Walk w;
        I3 mp = floor_i3_fast(rp);
        Found cur;
        if (eye && eye->first_valid) {  // wave-uniform
            w.node = eye->first_node; w.s = eye->first_s; w.anode = eye->first_anode; w.as = eye->first_as; w.last = mp;
            const int t = (int)(eye->first_w1 >> 24);
            cur.w0 = eye->first_w0;
            cur.w1 = eye->first_w1 & 0x00ffffffu;
            cur.plane = I3{((mp.x >> t) + dpos.x) << t, ((mp.y >> t) + dpos.y) << t, ((mp.z >> t) + dpos.z) << t};
        } else {
            reset(w);
            cur = find_checked(a, c, mp, dpos, w);
        }
        uint32_t cur_b = cur.w1 & 0xffu;  // medium byte: every Found carries 0 here when alpha == 0
        Axis ax{false, false};
        bool hit = false, go;
        uint32_t pw0 = 0u, pw1 = 0u;
        int i = 0;
        do {
            ax = dda_step(rp, dir, inv, push, cur.plane);
            mp = floor_i3_fast(rp);
            const bool inw = in_world_u(a, mp);
            if (inw) {
                pw0 = cur.w0; pw1 = cur.w1;
                const uint32_t prev_b = cur_b ? cur_b : iof_byte;
                cur = find_node(a, c, mp, dpos, w);
                cur_b = cur.w1 & 0xffu;
                hit = (cur_b ? cur_b : 85u) != prev_b;
            }
            ++i;
            go = inw && !hit && i < 1024;
        } while (go);
        const int axis = ax.x ? 0 : (ax.y ? 1 : 2);
        const float n = -comp(sd, axis);
        h.axis = axis; h.n = n;
        h.map = mp; h.point = rp; h.p0 = pw0; h.p1 = pw1; h.h0 = cur.w0; h.h1 = cur.w1;
        h.r_node = w.node; h.r_s = w.s; h.r_anode = w.anode; h.r_as = w.as; h.r_last = w.last;
        return hit;
    }

    // notInShadow (comp:333-377); the light direction is used as given
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, F3 ld, const Hit &h) {
        F3 rp = origin, inv;
        inv.x = (__builtin_fabsf(ld.x) < 1e-8f) ? 1e20f : 1.0f / ld.x;
        inv.y = (__builtin_fabsf(ld.y) < 1e-8f) ? 1e20f : 1.0f / ld.y;
        inv.z = (__builtin_fabsf(ld.z) < 1e-8f) ? 1e20f : 1.0f / ld.z;
        const I3 dpos{ld.x > 0.0f ? 1 : 0, ld.y > 0.0f ? 1 : 0, ld.z > 0.0f ? 1 : 0};
        const F3 push{sign_c(ld.x) * 0.001f, sign_c(ld.y) * 0.001f, sign_c(ld.z) * 0.001f};
        I3 mp = floor_i3_fast(rp);
        Walk w;  // resume where the primary ray stopped: the origin is 2e-3 off its hit point
        w.node = h.r_node; w.s = h.r_s; w.anode = h.r_anode; w.as = h.r_as; w.last = h.r_last;
        Found v = find_checked(a, c, mp, dpos, w);
        int lit = 1, i = 0;
        bool go;
        do {
            // occluder: alpha > 0.1 <=> alpha byte >= 26; illumination byte == 0 (comp:355)
            const bool occluder = (v.w0 >> 24) >= 26u && ((v.w1 >> 8) & 0xffu) == 0u;
            lit = occluder ? 0 : lit;
            (void)dda_step(rp, ld, inv, push, v.plane);
            mp = floor_i3_fast(rp);
            ++i;
            go = !occluder && in_world_u(a, mp) && i < 64;
            if (go) v = find_node(a, c, mp, dpos, w);
        } while (go);
        return lit;
    }
};

}  // namespace v3
}  // namespace vrt
