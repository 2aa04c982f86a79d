// vrt_kernels_v4.hip.h -- traversal variant "v4": the wide-node lookup of v3, with the march loop rebuilt around what
// a gfx950 SIMD charges per instruction.
//
// Measured (tools/micro/valu_rate, profiles/r02_valu_rate.txt; SIMD ticks per wave-instruction with 6-8 waves resident):
//   v_add/sub/mul_f32 0.6-0.9 | v_floor/fract 1.0-1.3 | v_fma_f32 1.2-1.4 | v_and/xor/add/sub_u32, v_lshrrev (vgpr) 1.4-1.65
//   | v_min/max_f32, v_min3 1.5-2.0 | v_mov 1.6-1.8 | v_cmp 2.0-2.7 | EVERY scalar instruction 2.0-2.7 | v_cndmask 2.7-2.9
//   | VOP3 integer (bfe, lshl_or, or3, add3, add_lshl, mad_u24), shifts by a constant, v_pk_*_f32, v_cvt_* 2.6-3.0
//   | a taken branch 3-3.5 (18 ticks of the wave's own time).
// The v3 kernel spends 2,287 instructions per wave at a mean of 2.2 ticks: 30 % of them scalar mask bookkeeping for
// divergent exits and nested loops, 14 v_mov per step for loop-carried values, packed f32 arithmetic that costs more
// than the two plain instructions it replaces, integer shifts to rebuild the node planes. Its SIMD issue time equals
// the launch time: the kernel is issue bound, so the remedy is a cheaper instruction stream, not more waves.
//
// What v4 changes (outputs identical: the lookup still returns octreeFind's node, the DDA arithmetic is untouched):
//   * the exit axis, the hit flag and the ray status live in vector registers, so leaving the loop needs one mask
//     operation instead of a merge per flag; one conditional block per step (the lookup that left its anchor);
//   * the world-bounds test (comp:224-226) moves into that block: a point inside the ray's current wide node or its
//     anchor is inside the world by construction;
//   * "no current node" is a walk state whose tests cannot pass (cell shift 0 and a `last` point no representable
//     floor() can come within 4 of), not an extra flag;
//   * planes, min-axis selection and the push in plain f32 / integer ops chosen from the table above.
#pragma once
#include "vrt_kernels_wide.hip.h"

namespace vrt {
namespace v4 {

using v3::floor_i3_fast;
using v3::in_world_u;
using v3::kAnchorShift;

struct Walk {                   // per-ray lookup state carried from one find to the next
    uint32_t node, cs;          // current wide node and log2 of its CELL side (node side = 4 cells)
    uint32_t anode, acs;        // anchor wide node (an ancestor of `node`, or `node` itself)
    I3 last;                    // the previous query point (inside both)
};

struct Found {
    uint32_t w0, w1;            // leaf words, or 0/0 for empty space
    I3 plane;                   // per axis: the face of the node found that a ray with signs `dpos` leaves through
};

enum : int { kGo = 0, kDone = 1, kOutside = 2 };

struct Trav {
    static constexpr bool kStagesLds = false;
    using Ctx = v3::Trav::Ctx;

    template <int BLOCK>
    static VRT_DEV void block_init(const KArgs &a, uint2 *, Ctx &c) { c.root = a.nodes[0]; }

    // No current node: cell shift 0 makes both tests "d < 4", and no floor() of a float can come that close to this
    // point (in the world every coordinate is sign-extended from bit 11; beyond 2^24 floor() is a multiple of 128).
    static VRT_DEV void reset(Walk &w) {
        w.node = w.anode = 0u; w.cs = w.acs = 0u;
        w.last = I3{0x55555555, 0x2aaaaaaa, 0x55555555};
    }

    // octreeFind (comp:137-220): the deepest octree node containing p, through the wide layout.
    // kGo: the answer came from a cell (f.w0, f.w1, f.plane set). kDone: from the record walk above the wide roots
    // (f set). kOutside: p is outside the world (comp:143-145): f.w0 / f.w1 untouched, f.plane not set.
    //
    // A lookup that has left its anchor (or has none) takes the one conditional block of the march loop: world-bounds
    // test, then wide root 0 when the point lies in its cube -- where the descent from the octree root would arrive
    // anyway -- else the record walk of v3 (the other seven octants of the reference's world: a nested, rarely
    // entered block). Inside the block everything is a select, so its lanes meet again after a handful of instructions.
    static VRT_DEV int find(const KArgs &a, const Ctx &c, I3 p, I3 dpos, Walk &w, Found &f) {
        const uint32_t d = (uint32_t)((p.x ^ w.last.x) | (p.y ^ w.last.y) | (p.z ^ w.last.z));
        const bool in_node = (d >> w.cs) < 4u;
        const bool in_anchor = (d >> w.acs) < 4u;     // the anchor contains the node: in_node implies in_anchor
        uint32_t node = in_node ? w.node : w.anode;
        uint32_t cs = in_node ? w.cs : w.acs;
        uint32_t anode = w.anode, acs = w.acs;
        int status = kGo;
        if (!in_anchor) {
            const int rs = a.root0_shift;
            const uint32_t out0 = (uint32_t)((p.x ^ a.root0_min[0]) | (p.y ^ a.root0_min[1]) | (p.z ^ a.root0_min[2])) >> (rs & 31);
            const bool in0 = a.n_roots != 0u && out0 == 0u;
            node = anode = a.root0_node;
            cs = acs = (uint32_t)(rs - 2);
            status = in_world_u(a, p) ? (in0 ? kGo : kDone) : kOutside;
            asm volatile("" : "+v"(status));   // a vector register, not a pair of lane masks to merge
            if (status == kDone) {   // in the world, outside wide root 0: walk the records (v3)
                v3::Walk w3;
                v3::Found f3;
                f3.w0 = 0u; f3.w1 = 0u; f3.plane = I3{0, 0, 0};
                uint32_t n3 = 0u;
                int s3 = 2;
                if (v3::Trav::descend_generic(a, c, p, dpos, w3, f3, n3, s3)) {
                    f.w0 = f3.w0; f.w1 = f3.w1; f.plane = f3.plane;
                    node = anode = 0u; cs = acs = 0u;                  // reset(): no current node
                    p = I3{0x55555555, 0x2aaaaaaa, 0x55555555};        // becomes w.last below
                } else {
                    node = anode = n3; cs = acs = (uint32_t)(s3 - 2);
                    status = kGo;
                }
            }
        }
        uint2 cell = make_uint2(0u, 0u);
        if (status == kGo) {
            for (;;) {
                const uint32_t ci = (__builtin_amdgcn_ubfe((uint32_t)p.x, cs, 2u) << 4) |
                                    (__builtin_amdgcn_ubfe((uint32_t)p.y, cs, 2u) << 2) |
                                    __builtin_amdgcn_ubfe((uint32_t)p.z, cs, 2u);
                cell = a.cells[(node << 6) | ci];
                if ((int)cell.y >= 0) break;      // bit 31: subdivided further, cell.x = child wide node
                node = cell.x;
                const bool up = cs == (uint32_t)kAnchorShift;    // the child has side 2^kAnchorShift: the new anchor
                cs -= 2u;
                anode = up ? node : anode;
                acs = up ? cs : acs;
            }
            const int t = (int)(cell.y >> 24);  // log2 side of the octree node found
            f.w0 = cell.x;
            f.w1 = cell.y & 0x00ffffffu;
            f.plane = I3{((p.x >> t) + dpos.x) << t, ((p.y >> t) + dpos.y) << t, ((p.z >> t) + dpos.z) << t};
        }
        w.node = node; w.cs = cs; w.anode = anode; w.acs = acs; w.last = p;
        return status;
    }

    // One DDA step (comp:278-307). The exit axis (comp:292): x when tx < ty && tx < tz, else y when ty < tz, else z;
    // with m = (ty < tz ? ty : tz) the first test is tx < m, and tStep = min(tx, min(ty, tz)) is the tMax of that axis.
    static VRT_DEV int dda_step(F3 &rp, F3 dir, F3 inv, F3 push, I3 plane) {
        const float tx = ((float)plane.x - rp.x) * inv.x;
        const float ty = ((float)plane.y - rp.y) * inv.y;
        const float tz = ((float)plane.z - rp.z) * inv.z;
        const bool yz = ty < tz;
        const float m = yz ? ty : tz;
        const bool ax = tx < m;
        const float t = ax ? tx : m;
        const int axis = ax ? 0 : (yz ? 1 : 2);
        const float rx = rp.x + dir.x * t, ry = rp.y + dir.y * t, rz = rp.z + dir.z * t;
        rp.x = axis == 0 ? rx + push.x : rx;
        rp.y = axis == 1 ? ry + push.y : ry;
        rp.z = axis == 2 ? rz + push.z : rz;
        return axis;
    }

    // hitMarching (comp:248-330)
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
        cur.w0 = 0u; cur.w1 = 0u; cur.plane = I3{0, 0, 0};
        if (eye && eye->first_valid) {  // wave-uniform: the first lookup was made by the host
            w.node = eye->first_node; w.cs = (uint32_t)(eye->first_s - 2);
            w.anode = eye->first_anode; w.acs = (uint32_t)(eye->first_as - 2); w.last = mp;
            const int t = (int)(eye->first_w1 >> 24);
            cur.w0 = eye->first_w0;
            cur.w1 = eye->first_w1 & 0x00ffffffu;
            cur.plane = I3{((mp.x >> t) + dpos.x) << t, ((mp.y >> t) + dpos.y) << t, ((mp.z >> t) + dpos.z) << t};
        } else {
            reset(w);
            if (find(a, c, mp, dpos, w, cur) == kOutside)   // comp:143-145, convention C8: zeroed data, the world's bounds
                cur.plane = I3{dpos.x ? a.wmax[0] : a.wmin[0], dpos.y ? a.wmax[1] : a.wmin[1], dpos.z ? a.wmax[2] : a.wmin[2]};
        }
        uint32_t cur_b = cur.w1 & 0xffu;  // medium byte: every Found carries 0 here when alpha == 0
        int axis = 2;
        uint32_t pw0 = 0u, pw1 = 0u;
        int i = 0;
        int hit = 0;
        bool go;
        do {
            axis = dda_step(rp, dir, inv, push, cur.plane);
            // the exit axis and the hit flag live in vector registers: as lane masks they would have to be merged into
            // the masks of the lanes that have already left the loop on every iteration (three scalar instructions each)
            asm volatile("" : "+v"(axis));
            mp = floor_i3_fast(rp);
            const uint32_t prev_b = cur_b ? cur_b : iof_byte;
            // a ray that leaves the world misses (comp:307-310): nothing reads its voxel words afterwards, so they
            // are updated unconditionally rather than through a select per register
            pw0 = cur.w0; pw1 = cur.w1;
            const bool inw = find(a, c, mp, dpos, w, cur) != kOutside;
            cur_b = cur.w1 & 0xffu;
            hit = (inw && (cur_b ? cur_b : 85u) != prev_b) ? 1 : 0;
            asm volatile("" : "+v"(hit));
            ++i;
            go = inw && hit == 0 && i < 1024;
        } while (go);
        const float n = -comp(sd, axis);
        h.axis = axis; h.n = n;
        h.map = mp; h.point = rp; h.p0 = pw0; h.p1 = pw1; h.h0 = cur.w0; h.h1 = cur.w1;
        h.r_node = w.node; h.r_s = (int)w.cs; h.r_anode = w.anode; h.r_as = (int)w.acs; h.r_last = w.last;
        asm volatile("" : "+v"(hit));   // re-read after the loop: otherwise the in-loop comparison is carried out as a merged lane mask
        return hit != 0;
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
        w.node = h.r_node; w.cs = (uint32_t)h.r_s; w.anode = h.r_anode; w.acs = (uint32_t)h.r_as; w.last = h.r_last;
        Found v;
        v.w0 = 0u; v.w1 = 0u; v.plane = I3{0, 0, 0};
        if (find(a, c, mp, dpos, w, v) == kOutside)
            v.plane = I3{dpos.x ? a.wmax[0] : a.wmin[0], dpos.y ? a.wmax[1] : a.wmin[1], dpos.z ? a.wmax[2] : a.wmin[2]};
        int lit = 1, i = 0;
        bool go;
        do {
            // occluder: alpha > 0.1 <=> alpha byte >= 26; illumination byte == 0 (comp:355)
            const bool occluder = (v.w0 >> 24) >= 26u && ((v.w1 >> 8) & 0xffu) == 0u;
            lit = occluder ? 0 : lit;
            (void)dda_step(rp, ld, inv, push, v.plane);
            mp = floor_i3_fast(rp);
            ++i;
            go = !occluder && i < 64;
            if (go) go = find(a, c, mp, dpos, w, v) != kOutside;
        } while (go);
        return lit;
    }
};

}  // namespace v4
}  // namespace vrt
