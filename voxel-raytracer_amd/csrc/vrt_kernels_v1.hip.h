// vrt_kernels_v1.hip.h -- traversal variant "v1": the straightforward form.
// octreeFind (comp:137-220) with explicit integer AABBs per level and a single
// cached parent, restarting at the root whenever the point leaves it -- the
// shader's own strategy over the 8-byte record layout. Kept as the measured
// baseline the tuned traversal (vrt_kernels.hip.h) is compared against.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace v1 {

struct Parent {            // the cached internal node (octreeFind's inout state, comp:137)
    uint32_t masks, base;  // w0 = child_mask | leaf_mask << 8, w1 = index of first child record
    I3 mn, mx;
};
struct Found {             // deepest node containing the query point
    uint32_t w0, w1;       // leaf words, or 0/0 when the point is in empty space
    I3 mn, mx;             // that node's AABB (half-open)
};

template <bool USE_LDS>
struct Trav {
    static constexpr bool kStagesLds = USE_LDS;
    struct Ctx {
        const uint2 *lds;
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

    static VRT_DEV Parent root_parent(const KArgs &a, const Ctx &c) {
        Parent p;
        p.masks = c.root.x; p.base = c.root.y;
        p.mn = I3{a.wmin[0], a.wmin[1], a.wmin[2]};
        p.mx = I3{a.wmax[0], a.wmax[1], a.wmax[2]};
        return p;
    }

    // On return `par` is the parent of the node found, as in the shader.
    static VRT_DEV Found find_node(const KArgs &a, const Ctx &c, I3 p, Parent &par) {
        Found f;
        f.w0 = 0u; f.w1 = 0u;
        if (!in_world(a, p)) {  // comp:143-145 (bounds are implementation-undefined there; convention C8)
            f.mn = I3{a.wmin[0], a.wmin[1], a.wmin[2]};
            f.mx = I3{a.wmax[0], a.wmax[1], a.wmax[2]};
            return f;
        }
        bool inside = p.x >= par.mn.x && p.y >= par.mn.y && p.z >= par.mn.z &&
                      p.x < par.mx.x && p.y < par.mx.y && p.z < par.mx.z;
        if (!inside) par = root_parent(a, c);
        for (int i = 0; i < 16; ++i) {
            int mx_ = par.mn.x + ((par.mx.x - par.mn.x) >> 1);  // extents are >= 0: >>1 == /2
            int my_ = par.mn.y + ((par.mx.y - par.mn.y) >> 1);
            int mz_ = par.mn.z + ((par.mx.z - par.mn.z) >> 1);
            bool hx = p.x >= mx_, hy = p.y >= my_, hz = p.z >= mz_;
            uint32_t ci = (hx ? 4u : 0u) | (hy ? 2u : 0u) | (hz ? 1u : 0u);
            I3 cmn{hx ? mx_ : par.mn.x, hy ? my_ : par.mn.y, hz ? mz_ : par.mn.z};
            I3 cmx{hx ? par.mx.x : mx_, hy ? par.mx.y : my_, hz ? par.mx.z : mz_};
            uint32_t bit = 1u << ci;
            if (!(par.masks & bit)) { f.mn = cmn; f.mx = cmx; return f; }  // absent child: empty space
            uint32_t idx = par.base + (uint32_t)__builtin_popcount(par.masks & 0xffu & (bit - 1u));
            uint2 rec = load_record(a, c, idx);
            if (par.masks & (bit << 8)) { f.w0 = rec.x; f.w1 = rec.y; f.mn = cmn; f.mx = cmx; return f; }
            par.masks = rec.x; par.base = rec.y; par.mn = cmn; par.mx = cmx;
        }
        f.mn = par.mn; f.mx = par.mx;  // unreachable for trees the uploader accepts (depth <= 15)
        return f;
    }

    // hitMarching (comp:248-330)
    static VRT_DEV bool march(const KArgs &a, const Ctx &c, F3 origin, F3 dir, float ray_iof, uint32_t /*iof_byte*/, Hit &h,
                              const View * = nullptr) {
        F3 rp = origin;
        float inv_len = 1.0f / __builtin_sqrtf(dot3(dir, dir));
        dir = scale3(dir, inv_len);
        F3 inv;
        inv.x = (__builtin_fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
        inv.y = (__builtin_fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
        inv.z = (__builtin_fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
        Parent par = root_parent(a, c);
        I3 mp = floor_i3(rp);
        Found cur = find_node(a, c, mp, par);
        // medium the ray is currently in: refraction index if (alpha > 0 && props[0] > 0)
        float cur_ref = refraction_of(cur.w1);
        bool cur_solid = (cur.w0 >> 24) != 0u && cur_ref > 0.0f;
        h.axis = 0; h.n = 0.0f;
        for (int i = 0; i < 1024; ++i) {
            F3 tp;
            tp.x = (dir.x > 0.0f ? (float)cur.mx.x : (float)cur.mn.x) - rp.x;
            tp.y = (dir.y > 0.0f ? (float)cur.mx.y : (float)cur.mn.y) - rp.y;
            tp.z = (dir.z > 0.0f ? (float)cur.mx.z : (float)cur.mn.z) - rp.z;
            float tx = tp.x * inv.x, ty = tp.y * inv.y, tz = tp.z * inv.z;
            float t = fmin_c(tx, fmin_c(ty, tz));
            int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
            float sd = sign_c(comp(dir, axis));
            float n = -sd;
            h.axis = axis; h.n = n;
            rp.x = rp.x + dir.x * t; rp.y = rp.y + dir.y * t; rp.z = rp.z + dir.z * t;
            float push = sd * 0.0001f;
            if (axis == 0) rp.x = rp.x + push; else if (axis == 1) rp.y = rp.y + push; else rp.z = rp.z + push;
            mp = floor_i3(rp);
            if (!in_world(a, mp)) return false;
            uint32_t pw0 = cur.w0, pw1 = cur.w1;
            float prev_ref = cur_solid ? cur_ref : ray_iof;
            cur = find_node(a, c, mp, par);
            cur_ref = refraction_of(cur.w1);
            cur_solid = (cur.w0 >> 24) != 0u && cur_ref > 0.0f;
            float now_ref = cur_solid ? cur_ref : 1.0f;
            if (__builtin_fabsf(now_ref - prev_ref) > 0.0001f) {
                h.map = mp; h.point = rp; h.p0 = pw0; h.p1 = pw1; h.h0 = cur.w0; h.h1 = cur.w1;
                return true;
            }
        }
        return false;
    }

    // notInShadow (comp:333-377); the light direction is used as given
    static VRT_DEV int shadow(const KArgs &a, const Ctx &c, F3 origin, F3 ld, const Hit & /*resume hint unused*/) {
        F3 rp = origin, inv;
        inv.x = (__builtin_fabsf(ld.x) < 1e-8f) ? 1e20f : 1.0f / ld.x;
        inv.y = (__builtin_fabsf(ld.y) < 1e-8f) ? 1e20f : 1.0f / ld.y;
        inv.z = (__builtin_fabsf(ld.z) < 1e-8f) ? 1e20f : 1.0f / ld.z;
        I3 mp = floor_i3(rp);
        Parent par = root_parent(a, c);
        for (int i = 0; i < 64; ++i) {
            Found v = find_node(a, c, mp, par);
            float alpha = (float)(v.w0 >> 24) / 255.0f;
            // properties[1] == 0  <=>  illumination byte == 0
            if (alpha > 0.1f && ((v.w1 >> 8) & 0xffu) == 0u) return 0;
            F3 tp;
            tp.x = (ld.x > 0.0f ? (float)v.mx.x : (float)v.mn.x) - rp.x;
            tp.y = (ld.y > 0.0f ? (float)v.mx.y : (float)v.mn.y) - rp.y;
            tp.z = (ld.z > 0.0f ? (float)v.mx.z : (float)v.mn.z) - rp.z;
            float tx = tp.x * inv.x, ty = tp.y * inv.y, tz = tp.z * inv.z;
            float t = fmin_c(tx, fmin_c(ty, tz));
            int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
            rp.x = rp.x + ld.x * t; rp.y = rp.y + ld.y * t; rp.z = rp.z + ld.z * t;
            float push = sign_c(comp(ld, axis)) * 0.001f;
            if (axis == 0) rp.x = rp.x + push; else if (axis == 1) rp.y = rp.y + push; else rp.z = rp.z + push;
            mp = floor_i3(rp);
            if (!in_world(a, mp)) return 1;
        }
        return 1;
    }
};

}  // namespace v1
}  // namespace vrt
