// vrt_kernels_v4.hip.h -- traversal variant "v4": the wide-node lookup of v3, with the march loop rebuilt around what
// a gfx950 SIMD charges per instruction.
//
// Measured (tools/micro/valu_rate, profiles/r02_valu_rate.txt; SIMD ticks per wave64 instruction by the launch's wall
// time with 6-8 waves resident -- the third column; the per-wave s_memtime columns under-read because the probe's own
// registers limit how many of its waves are resident):
//   2.5-2.6  v_add/sub/mul/fma_f32, v_and/xor/add/sub_u32, v_lshrrev by a VGPR amount, v_mov
//   4.3-4.6  v_cmp, v_cndmask, v_floor/fract, v_min/max/med3, v_cvt_*, shifts by a constant, every VOP3 integer op (bfe,
//            lshl_or, or3, add3, lshl_add, mad_u24, perm, bfi), v_pk_*_f32, v_readfirstlane; every scalar instruction and
//            a branch not taken cost the same on the scalar port, which overlaps with the vector one (v_add + s_add: 2.65
//            per pair)
//   8.4      v_rcp / v_sqrt / v_rsq
// The v3 kernel spends 2,287 instructions per wave: 30 % of them scalar mask bookkeeping for divergent exits and nested
// loops, 14 v_mov per step for loop-carried values, packed f32 arithmetic that costs as much as the two plain
// instructions it replaces and more than one, integer shifts and conversions to rebuild the node planes. Its vector
// issue time is within 20 % of the launch time: the kernel is issue bound, so the remedy is a cheaper instruction
// stream, not more waves.
//
// What v4 changes (outputs identical: the lookup still returns octreeFind's node, the DDA arithmetic is untouched):
//   * the exit axis, the hit flag and the ray status live in vector registers, so leaving the loop needs one mask
//     operation instead of a merge per flag; one conditional block per step (the lookup that left its anchor);
//   * the world-bounds test (comp:224-226) moves into that block: a point inside the ray's current wide node or its
//     anchor is inside the world by construction;
//   * "no current node" is a walk state whose tests cannot pass (cell shift 0 and a `last` point no representable
//     floor() can come within 4 of), not an extra flag;
//   * planes, min-axis selection and the push in plain f32 / integer ops from the first row of the table above: the cells are read
//     in a second form (vrt_layout.h to_cell4) that carries 2^t as a float exponent, so a node's planes are
//     (floor(floor(p) * 2^-t) + dpos) * 2^t in four f32 operations per axis -- every step exact, hence the same floats
//     the integer arithmetic gives -- the child node as a byte offset, and the medium byte with 85 for empty space, so
//     that "the medium changed" is one comparison;
//   * the first cell load of a lookup is straight-line code, only a further descent loops; the walk's reference point
//     moves only when the node changes.
#pragma once
#include "vrt_kernels_wide.hip.h"

namespace vrt {
namespace v4 {

using v3::in_world_u;
using v3::kAnchorShift;

struct Walk {                   // per-ray lookup state carried from one find to the next
    uint32_t node, cs;          // current wide node (BYTE offset in cells4) and log2 of its CELL side (node side = 4 cells)
    uint32_t anode, acs;        // anchor wide node (an ancestor of `node`, or `node` itself)
    I3 last;                    // a point inside `node` (hence inside the anchor)
};

struct Found {
    uint32_t x, y;              // the cell in to_cell4() form: leaf word 0; medium byte | illum << 8 | k ... (vrt_layout.h)
    F3 plane;                   // per axis: the face of the node found that a ray with signs `dpos` leaves through
};

enum : int { kGo = 0, kDone = 1, kOutside = 2 };
constexpr uint32_t kExpMask = 0x0f800000u;   // (t + 1) << 23; 0 in a subdivided cell

// leaf word 1 as the other traversals report it: refr | illum << 8 | k << 16, refr 0 when alpha is 0
VRT_DEV uint32_t word1_of(uint32_t x, uint32_t y) {
    const uint32_t refr = ((x >> 24) == 0u || (y & (1u << 29)) != 0u) ? 0u : (y & 0xffu);
    return refr | (y & 0x007fff00u) | ((y >> 5) & 0x00800000u);
}
// to_cell4() of a leaf given as (word 0, word 1 with refr 0 under alpha 0), t + 1 = tp1
VRT_DEV uint32_t cell4_y(uint32_t w0, uint32_t w1, uint32_t tp1) {
    const uint32_t b = w1 & 0xffu;
    const uint32_t raw0 = (b == 0u && (w0 >> 24) != 0u) ? 1u : 0u;
    return (b ? b : 85u) | (w1 & 0x007fff00u) | ((w1 & 0x00800000u) << 5) | (tp1 << 23) | (raw0 << 29);
}

// EYE85: the caller guarantees that every march() starts in refraction byte 85 (1.0) -- the primary rays of views whose
// eye is in empty space, which the dispatcher checks (it has the eye's voxel) -- so the kernel holds one march loop and
// its registers; otherwise march() runs the loop that takes a ray's own starting medium (the full path tracer).
template <bool EYE85>
struct TravT {
    static constexpr bool kStagesLds = false;
    using Ctx = v3::Trav::Ctx;
    using Eye85 = TravT<true>;   // the same traversal with the one march loop of rays that start in empty space
    using General = TravT<false>;   // ... and with the loop that takes a ray's own starting medium

    template <int BLOCK>
    static VRT_DEV void block_init(const KArgs &a, uint2 *, Ctx &c) { c.root = a.nodes[0]; }

    // No current node: cell shift 0 makes both tests "d < 4", and no floor() of a float can come that close to this
    // point (in the world every coordinate is sign-extended from bit 11; beyond 2^24 floor() is a multiple of 128).
    static VRT_DEV void reset(Walk &w) {
        w.node = w.anode = 0u; w.cs = w.acs = 0u;
        w.last = I3{0x55555555, 0x2aaaaaaa, 0x55555555};
    }

    static VRT_DEV uint2 load_cell(const KArgs &a, uint32_t node, uint32_t cs, I3 p) {
        const uint32_t bx = __builtin_amdgcn_ubfe((uint32_t)p.x, cs, 2u), by = __builtin_amdgcn_ubfe((uint32_t)p.y, cs, 2u),
                       bz = __builtin_amdgcn_ubfe((uint32_t)p.z, cs, 2u);
        // node + (((bx << 2 | by) << 2 | bz) << 3) as three v_lshl_add_u32 (the compiler spreads it over four instructions)
        uint32_t off;
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(bx), "v"(by));
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(off), "v"(bz));
        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(off) : "v"(off), "v"(node));
        return *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(a.cells4) + off);
    }

    // octreeFind (comp:137-220): the deepest octree node containing p (= floor of a ray position, pf the same as
    // floats), through the wide layout.
    // kGo: the answer came from a cell (f set). kDone: from the record walk above the wide roots (f set).
    // kOutside: p is outside the world (comp:143-145): f untouched -- and w then refers to p as if it were a point of wide
    // root 0: a caller that goes on after kOutside (only the first lookups of march() and shadow() do) must reset(w).
    //
    // A lookup that has left its anchor (or has none) takes the one conditional block of the march loop: world-bounds
    // test, then wide root 0 when the point lies in its cube -- where the descent from the octree root would arrive
    // anyway -- else the record walk of v3 (the other seven octants of the reference's world: a nested, rarely
    // entered block). Inside the block everything is a select, so its lanes meet again after a handful of instructions.
    static VRT_DEV int find(const KArgs &a, const Ctx &c, I3 p, F3 pf, I3 dpos, F3 dposf, Walk &w, Found &f, bool forward, uint32_t leaving_m = 85u) {
        const uint32_t d = (uint32_t)((p.x ^ w.last.x) | (p.y ^ w.last.y) | (p.z ^ w.last.z));
        const bool in_node = (d >> w.cs) < 4u;
        const bool in_anchor = (d >> w.acs) < 4u;     // the anchor contains the node: in_node implies in_anchor
        uint32_t node = in_node ? w.node : w.anode;
        uint32_t cs = in_node ? w.cs : w.acs;
        uint32_t anode = w.anode, acs = w.acs;
        I3 last = w.last;
        int status = kGo;
        if (!in_anchor) {
            const int rs = a.root0_shift;
            const uint32_t out0 = (uint32_t)((p.x ^ a.root0_min[0]) | (p.y ^ a.root0_min[1]) | (p.z ^ a.root0_min[2])) >> (rs & 31);
            const bool in0 = a.n_roots != 0u && out0 == 0u;
            node = anode = a.root0_node << 9;
            cs = acs = (uint32_t)(rs - 2);
            last = p;
            // KArgs::root0_only: the tree has nothing outside wide root 0 (the dispatcher checked the records above it). A ray
            // that HAS BEEN inside that cube -- its walk holds a node: acs != 0 -- and is outside it now cannot come back
            // and has only empty space ahead: it misses whatever it still crosses, so the lookup says "outside" at once
            // instead of walking the records of the empty octants (1.5 such walks per wave of the bench frame, 125 vector
            // and 153 scalar instructions each). "Cannot come back": every step adds t * dir and a push of dir's sign, so
            // each coordinate moves one way only, PROVIDED t >= 0 -- true unless a component of dir lies in (-1e-8, 0],
            // whose reciprocal the shader replaces by +1e20 (comp:256-258) while its plane is the cell's near face: such a
            // ray steps backwards. `forward` (forward_only()) excludes those; they and rays not yet inside take the walk.
            // And "misses": empty space is no event only for a ray that is IN empty space -- for one inside a solid or glass
            // (leaving_m, the medium byte its hit test compares with, is not 85) the empty octant is the hit (found by the
            // parity fuzz: eyes inside the models' bases, rays leaving through the floor of the cube).
            const bool gone = a.root0_only != 0 && forward && w.acs != 0u && leaving_m == 85u;
            status = kGo;
            if (!in0) status = gone ? kOutside : (in_world_u(a, p) ? kDone : kOutside);
            asm volatile("" : "+v"(status));   // a vector register, not a pair of lane masks to merge
            if (status == kDone) {   // in the world, outside wide root 0: walk the records (v3)
                v3::Walk w3;
                v3::Found f3;
                f3.w0 = 0u; f3.w1 = 0u; f3.plane = I3{0, 0, 0};
                uint32_t n3 = 0u;
                int s3 = 2;
                if (v3::Trav::descend_generic(a, c, p, dpos, w3, f3, n3, s3)) {
                    f.x = f3.w0; f.y = cell4_y(f3.w0, f3.w1, 1u);
                    f.plane = F3{(float)f3.plane.x, (float)f3.plane.y, (float)f3.plane.z};
                    node = anode = 0u; cs = acs = 0u;                  // reset(): no current node
                    last = I3{0x55555555, 0x2aaaaaaa, 0x55555555};
                } else {
                    node = anode = n3 << 9; cs = acs = (uint32_t)(s3 - 2);
                    status = kGo;
                }
            }
        }
        if (status == kGo) {
            uint2 cell = load_cell(a, node, cs, p);
            uint32_t e = cell.y & kExpMask;
            if (e == 0u) {            // subdivided further: cell.x = the child wide node
                do {
                    node = cell.x;
                    const bool up = cs == (uint32_t)kAnchorShift;    // the child has side 2^kAnchorShift: the new anchor
                    cs -= 2u;
                    anode = up ? node : anode;
                    acs = up ? cs : acs;
                    cell = load_cell(a, node, cs, p);
                    e = cell.y & kExpMask;
                } while (e == 0u);
                last = p;
            }
            f.x = cell.x; f.y = cell.y;
            // the node found has side 2^t, t + 1 in the exponent field: its planes are ((p >> t) + dpos) << t
            const float side = __uint_as_float(e + 0x3f000000u), inv_side = __uint_as_float(0x40000000u - e);
            f.plane.x = (__builtin_floorf(pf.x * inv_side) + dposf.x) * side;
            f.plane.y = (__builtin_floorf(pf.y * inv_side) + dposf.y) * side;
            f.plane.z = (__builtin_floorf(pf.z * inv_side) + dposf.z) * side;
        }
        w.node = node; w.cs = cs; w.anode = anode; w.acs = acs; w.last = last;
        return status;
    }

    // no component of the direction in (-1e-8, 0]: every step of the DDA has t >= 0 (see find())
    static VRT_DEV bool forward_only(F3 d) {
        return (d.x > 0.0f || d.x <= -1e-8f) && (d.y > 0.0f || d.y <= -1e-8f) && (d.z > 0.0f || d.z <= -1e-8f);
    }

    static VRT_DEV F3 world_planes(const KArgs &a, I3 dpos) {  // comp:143-145, convention C8: the world's bounds
        return F3{(float)(dpos.x ? a.wmax[0] : a.wmin[0]), (float)(dpos.y ? a.wmax[1] : a.wmin[1]), (float)(dpos.z ? a.wmax[2] : a.wmin[2])};
    }

    // One DDA step (comp:278-307). The exit axis (comp:292): x when tx < ty && tx < tz, else y when ty < tz, else z;
    // with m = (ty < tz ? ty : tz) the first test is tx < m, and tStep = min(tx, min(ty, tz)) is the tMax of that axis.
    // The vector port is what binds this kernel (a v_cndmask costs 4.4 ticks of it, a v_add_f32 2.6, scalar mask
    // arithmetic none): the push (comp:300-304) is added under the axis' lane mask instead of being selected per axis.
    // The two comparisons and everything that hangs on them are written out: the compiler computes each condition AND
    // its negation with a v_cmp of its own (four per step, 4.4 ticks each) and branches around every masked add. Here:
    // two v_cmp into scalar mask pairs, two v_cndmask on them, three adds under exec = the axis' lanes (an add with no
    // lane enabled costs its issue slot and nothing else). gfx950 needs two wait states between a VALU instruction that
    // writes a scalar register and a VALU instruction that reads it as a mask (s_nop 1); scalar instructions interlock.
    struct Axis { bool x, yz; };   // exit axis: x ? 0 : (yz ? 1 : 2)
    static VRT_DEV Axis dda_step(F3 &rp, F3 dir, F3 inv, F3 push, F3 plane) {
        const float tx = (plane.x - rp.x) * inv.x;
        const float ty = (plane.y - rp.y) * inv.y;
        const float tz = (plane.z - rp.z) * inv.z;
        uint64_t mx, myz;
        float t;
        asm("v_cmp_lt_f32_e64 %[myz], %[ty], %[tz]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[t], %[tz], %[ty], %[myz]\n\t"     // m = ty < tz ? ty : tz
            "v_cmp_lt_f32_e64 %[mx], %[tx], %[t]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32_e64 %[t], %[t], %[tx], %[mx]"             // t = tx < m ? tx : m
            : [t] "=&v"(t), [mx] "=&s"(mx), [myz] "=&s"(myz)
            : [tx] "v"(tx), [ty] "v"(ty), [tz] "v"(tz));
        float rx = rp.x + dir.x * t, ry = rp.y + dir.y * t, rz = rp.z + dir.z * t;
        uint64_t save, rest;
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     "s_mov_b64 exec, %[mx]\n\t"                    // a v_cmp result holds no lane that is not enabled
                     "v_add_f32_e32 %[rx], %[px], %[rx]\n\t"
                     "s_andn2_b64 %[rest], %[save], %[mx]\n\t"
                     "s_and_b64 exec, %[rest], %[myz]\n\t"
                     "v_add_f32_e32 %[ry], %[py], %[ry]\n\t"
                     "s_andn2_b64 exec, %[rest], %[myz]\n\t"
                     "v_add_f32_e32 %[rz], %[pz], %[rz]\n\t"
                     "s_mov_b64 exec, %[save]"
                     : [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz), [save] "=&s"(save), [rest] "=&s"(rest)
                     : [mx] "s"(mx), [myz] "s"(myz), [px] "v"(push.x), [py] "v"(push.y), [pz] "v"(push.z)
                     : "scc");
        rp.x = rx; rp.y = ry; rp.z = rz;
        // the masks as per-lane conditions again: no instruction, the compiler keeps such conditions as lane masks
        return Axis{__builtin_amdgcn_inverse_ballot_w64(mx), __builtin_amdgcn_inverse_ballot_w64(myz)};
    }

    static VRT_DEV void floor_both(F3 rp, F3 &pf, I3 &p) {
        pf = F3{__builtin_floorf(rp.x), __builtin_floorf(rp.y), __builtin_floorf(rp.z)};
        p = I3{(int)pf.x, (int)pf.y, (int)pf.z};   // v_cvt_i32_f32 of an integer-valued float: saturates, NaN -> 0, like v_cvt_flr_i32_f32
    }

    // The march loop. IOF85: every lane's starting medium is refraction byte 85 (1.0: the eye in empty space), so the
    // medium a step leaves is the mapped byte of the cell before (see to_cell4()); otherwise empty space counts as
    // the ray's own starting medium on the leaving side (comp:318-326).
    template <bool IOF85>
    static VRT_DEV bool march_loop(const KArgs &a, const Ctx &c, F3 &rp, F3 dir, F3 inv, F3 push, I3 dpos, F3 dposf, Walk &w, Found &cur,
                                   uint32_t iof_byte, int &axis, uint32_t &px, uint32_t &py, I3 &mp, bool forward, int *iters = nullptr) {
        Axis ax{false, false};
        F3 pf;
        int i = 0, status = kGo;
        uint32_t prev_m = 0u;
        bool go;
        do {
            ax = dda_step(rp, dir, inv, push, cur.plane);
            floor_both(rp, pf, mp);
            const uint32_t cur_m = cur.y & 0xffu;
            if constexpr (IOF85) prev_m = cur_m;
            else prev_m = ((cur.x >> 24) == 0u || (cur.y & (1u << 29)) != 0u) ? iof_byte : cur_m;
            // a ray that leaves the world misses (comp:307-310) and find() leaves `cur` alone then: nothing reads its voxel
            // words afterwards, so the previous voxel is updated unconditionally rather than through selects
            px = cur.x; py = cur.y;
            status = find(a, c, mp, pf, dpos, dposf, w, cur, forward, prev_m);
            asm volatile("" : "+v"(status));
            bool hit = (cur.y & 0xffu) != prev_m;
            if constexpr (!IOF85) hit = hit && status != kOutside;
            ++i;
            go = status != kOutside && !hit && i < 1024;
        } while (go);
        axis = ax.x ? 0 : (ax.yz ? 1 : 2);   // two lane masks merged per iteration by scalar instructions: off the vector port
        if (iters) *iters = i;
        // the hit flag from the registers the lane left the loop with (IOF85: outside the world `cur` is unchanged, so the
        // bytes are equal)
        asm volatile("" : "+v"(prev_m), "+v"(status));
        bool hit = (cur.y & 0xffu) != prev_m;
        if constexpr (!IOF85) hit = hit && status != kOutside;
        return hit;
    }

    // hitMarching (comp:248-330)
    static VRT_DEV bool march(const KArgs &a, const Ctx &c, F3 origin, F3 dir, float ray_iof, uint32_t iof_byte, Hit &h,
                              const View *eye = nullptr) {
        (void)ray_iof;
        F3 rp = origin;
        F3 inv;
        // hitMarching normalises the direction again and inverts it (comp:250-258). For the primary rays of a view whose
        // prologue ran in its in-range form (View::gen_fast: `dir` is then a unit vector up to rounding) these 1/x and sqrt
        // are in range too -- a component below 1e-8 is replaced, so the reciprocal only has to be right from there up.
        if (eye && eye->gen_fast) {
            dir = scale3(dir, rcp_inrange(sqrt_inrange(dot3(dir, dir))));
            inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : rcp_inrange(dir.x);
            inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : rcp_inrange(dir.y);
            inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : rcp_inrange(dir.z);
        } else {
            dir = scale3(dir, 1.0f / __builtin_sqrtf(dot3(dir, dir)));
            inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
            inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
            inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
        }
        const I3 dpos{dir.x > 0.0f ? 1 : 0, dir.y > 0.0f ? 1 : 0, dir.z > 0.0f ? 1 : 0};
        const F3 dposf{dir.x > 0.0f ? 1.0f : 0.0f, dir.y > 0.0f ? 1.0f : 0.0f, dir.z > 0.0f ? 1.0f : 0.0f};
        const F3 sd{sign_c(dir.x), sign_c(dir.y), sign_c(dir.z)};
        const F3 push{sd.x * 0.0001f, sd.y * 0.0001f, sd.z * 0.0001f};  // comp:300-304
        Walk w;
        F3 pf;
        I3 mp;
        floor_both(rp, pf, mp);
        Found cur;
        cur.x = 0u; cur.y = 85u | (1u << 23); cur.plane = F3{0.0f, 0.0f, 0.0f};   // empty space
        if (eye && eye->first_valid) {  // wave-uniform: the first lookup was made by the host
            w.node = eye->first_node << 9; w.cs = (uint32_t)(eye->first_s - 2);
            w.anode = eye->first_anode << 9; w.acs = (uint32_t)(eye->first_as - 2); w.last = mp;
            const uint32_t t = eye->first_w1 >> 24;
            cur.x = eye->first_w0;
            cur.y = cell4_y(eye->first_w0, eye->first_w1 & 0x00ffffffu, t + 1u);
            const float side = __uint_as_float((127u + t) << 23), inv_side = __uint_as_float((127u - t) << 23);
            cur.plane = F3{(__builtin_floorf(pf.x * inv_side) + dposf.x) * side, (__builtin_floorf(pf.y * inv_side) + dposf.y) * side,
                           (__builtin_floorf(pf.z * inv_side) + dposf.z) * side};
        } else {
            reset(w);
            // an eye outside the world: find() has noted that point as the walk's reference, and the next point -- still
            // outside, a few units on -- would pass for a point of wide root 0: no current node again
            if (find(a, c, mp, pf, dpos, dposf, w, cur, false) == kOutside) { cur.plane = world_planes(a, dpos); reset(w); }
        }
        int axis = 2;
        uint32_t px = 0u, py = 85u | (1u << 23);
        bool hit;
        // one loop per instantiation: choosing between the two per wave (all lanes in refraction 1.0 or not) made the full
        // path tracer hold both and spill 16 registers at its five waves per SIMD
        const bool forward = forward_only(dir);
#ifdef VRT_EXP_STATS
        if constexpr (EYE85) hit = march_loop<true>(a, c, rp, dir, inv, push, dpos, dposf, w, cur, iof_byte, axis, px, py, mp, forward, &h.iters);
        else hit = march_loop<false>(a, c, rp, dir, inv, push, dpos, dposf, w, cur, iof_byte, axis, px, py, mp, forward, &h.iters);
#else
        if constexpr (EYE85) hit = march_loop<true>(a, c, rp, dir, inv, push, dpos, dposf, w, cur, iof_byte, axis, px, py, mp, forward);
        else hit = march_loop<false>(a, c, rp, dir, inv, push, dpos, dposf, w, cur, iof_byte, axis, px, py, mp, forward);
#endif
        const float n = -comp(sd, axis);
        h.axis = axis; h.n = n;
        h.map = mp; h.point = rp;
        h.p0 = px; h.p1 = word1_of(px, py); h.h0 = cur.x; h.h1 = word1_of(cur.x, cur.y);
        h.r_node = w.node; h.r_s = (int)w.cs; h.r_anode = w.anode; h.r_as = (int)w.acs; h.r_last = w.last;
        return hit;
    }

    // notInShadow (comp:333-377); the light direction is used as given. Its reciprocal, push and signs are the same
    // for every ray of the launch: the dispatcher made them (KArgs::light_*, handed over in `ls`).
    static constexpr bool kHostLight = true;
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, const LightSetup &ls, const Hit &h) {
        F3 rp = origin;
        const F3 ld = ls.dir, inv = ls.inv, push = ls.push, dposf = ls.dposf;
        const I3 dpos = ls.dpos;
        F3 pf;
        I3 mp;
        floor_both(rp, pf, mp);
        Walk w;  // resume where the primary ray stopped: the origin is 2e-3 off its hit point
        w.node = h.r_node; w.cs = (uint32_t)h.r_s; w.anode = h.r_anode; w.acs = (uint32_t)h.r_as; w.last = h.r_last;
        Found v;
        v.x = 0u; v.y = 85u | (1u << 23); v.plane = F3{0.0f, 0.0f, 0.0f};
        const bool forward = forward_only(ld);
        // the first lookup never takes the root0_only shortcut: the walk it resumes is the primary ray's, and "has been inside
        // wide root 0" must be this ray's own history (an origin 2e-3 outside the cube's face starts outside)
        if (find(a, c, mp, pf, dpos, dposf, w, v, false) == kOutside) { v.plane = world_planes(a, dpos); reset(w); }   // as in march()
        int lit = 1, i = 0;
        bool go;
        do {
            // occluder: alpha > 0.1 <=> alpha byte >= 26; illumination byte == 0 (comp:355)
            const bool occluder = (v.x >> 24) >= 26u && (v.y & 0xff00u) == 0u;
            lit = occluder ? 0 : lit;
            (void)dda_step(rp, ld, inv, push, v.plane);
            floor_both(rp, pf, mp);
            ++i;
            go = !occluder && i < 64;
            if (go) go = find(a, c, mp, pf, dpos, dposf, w, v, forward) != kOutside;
        } while (go);
        return lit;
    }
    // the same from the direction alone (the full path tracer's call)
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, F3 ld, const Hit &h) {
        LightSetup ls;
        ls.dir = ld;
        ls.inv.x = (__builtin_fabsf(ld.x) < 1e-8f) ? 1e20f : 1.0f / ld.x;
        ls.inv.y = (__builtin_fabsf(ld.y) < 1e-8f) ? 1e20f : 1.0f / ld.y;
        ls.inv.z = (__builtin_fabsf(ld.z) < 1e-8f) ? 1e20f : 1.0f / ld.z;
        ls.dpos = I3{ld.x > 0.0f ? 1 : 0, ld.y > 0.0f ? 1 : 0, ld.z > 0.0f ? 1 : 0};
        ls.dposf = F3{ld.x > 0.0f ? 1.0f : 0.0f, ld.y > 0.0f ? 1.0f : 0.0f, ld.z > 0.0f ? 1.0f : 0.0f};
        ls.push = F3{sign_c(ld.x) * 0.001f, sign_c(ld.y) * 0.001f, sign_c(ld.z) * 0.001f};
        return shadow(a, c, origin, ls, h);
    }
};


using Trav = TravT<true>;       // modes 0, 1 when the eye is in empty space (else the dispatcher takes v3)
using TravAny = TravT<false>;   // the full path tracer

}  // namespace v4
}  // namespace vrt
