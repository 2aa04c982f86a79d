// vrt_full.hip.h -- VRT_MODE_FULL: the whole of pathTrace (shaders/raytracing.comp:435-622):
// the 8-deep ray stack with glass reflection/refraction (:546-572), Beer-Lambert absorption
// (:482-486, :512-516), emission and ambient terms (:574-594), the cosine-weighted diffuse bounce
// (:596-616) with the PCG-hash RNG (:379-417). The traversal (march / shadow) is the
// TRAV policy shared with the primary-ray kernels; this file is the shading state machine only.
//
// It mirrors the CPU restatement operation for operation (same dot/normalize/cross forms, same
// polynomial exp/log/sin/cos standing in for GLSL's implementation-defined transcendentals), so the
// outputs are bit-identical to it. The ray stack lives in private (scratch) memory: 8 x 68 bytes per lane.
#pragma once
#include "vrt_common.hip.h"

namespace vrt {
namespace full {

constexpr int kMaxRays = 8;          // MAX_RAYS  (comp:6)
constexpr int kBounces = 1;          // BOUNCES   (comp:8)
constexpr int kIndirectSamples = 1;  // INDIRECT_SAMPLES (comp:7)

struct RayS {            // comp:57-68 without the fields nothing reads (defined, the alpha of the two colours)
    F3 o, d;
    float iof, weight;
    float tint[3];
    float dim;           // distanceInMedium
    float mc[3];         // mediumColor.rgb
    float md;            // mediumDensity
    int depth;
};

// What waits on the stack: only the REFLECTED ray of a glass hit ever does (the refracted one and every diffuse bounce are marched
// next and take the finished ray's registers), and such a ray has depth 0 and travels in the medium of the voxel before the
// surface, so its medium colour and density are that voxel's colour word (comp:556-562: lastVoxel.color, lastVoxel.color.a * 5):
// 13 words instead of 17 per entry.
struct StackRay {
    F3 o, d;
    float iof, weight;
    float tint[3];
    float dim;
    uint32_t medium;     // leaf word 0 (R | G<<8 | B<<16 | alpha<<24) of the voxel the ray travels in
};

VRT_DEV F3 cross3(F3 x, F3 y) { return F3{x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }

VRT_DEV float det_logf(float x) {  // x > 0, normal (Cephes logf, plain mul/add)
    uint32_t u = __float_as_uint(x);
    int e = (int)(u >> 23) - 126;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m; }
    m = m - 1.0f;
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m + -1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m + -1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m + -1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m + -2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    float r = m + y;
    r = r + fe * 0.693359375f;
    return r;
}

VRT_DEV float det_powf(float x, float y) {
    if (x <= 0.0f) return 0.0f;
    if (x < 1.17549435e-38f) return 0.0f;
    return det_expf(y * det_logf(x));
}

VRT_DEV void det_sincos(float xin, float &s_out, float &c_out) {  // Cephes sinf/cosf octant reduction
    float x = __builtin_fabsf(xin);
    int sign_s = xin < 0.0f ? -1 : 1, sign_c = 1;
    int j = (int)(x * 1.27323954473516f);
    float y = (float)j;
    if (j & 1) { j += 1; y = y + 1.0f; }
    j &= 7;
    if (j > 3) { sign_s = -sign_s; sign_c = -sign_c; j -= 4; }
    if (j > 1) sign_c = -sign_c;
    x = x - y * 0.78515625f;
    x = x - y * 2.4187564849853515625e-4f;
    x = x - y * 3.77489497744594108e-8f;
    float z = x * x;
    float ps = -1.9515295891e-4f;
    ps = ps * z + 8.3321608736e-3f;
    ps = ps * z + -1.6666654611e-1f;
    ps = ps * z * x + x;
    float pc = 2.443315711809948e-5f;
    pc = pc * z + -1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    float sv, cv;
    if (j == 1 || j == 2) { sv = pc; cv = ps; } else { sv = ps; cv = pc; }
    s_out = sign_s < 0 ? -sv : sv;
    c_out = sign_c < 0 ? -cv : cv;
}

// comp:381-399
VRT_DEV uint32_t rng_init(int px, int py, int sample) {
    uint32_t seed = (uint32_t)px + (uint32_t)py * 1920u + 123456u + (uint32_t)sample * 78901u;
    uint32_t st = seed * 747796405u + 2891336453u;
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (w >> 22u) ^ w;
}
VRT_DEV float rng_next(uint32_t &state) {
    state = state * 747796405u + 2891336453u;
    uint32_t w = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    state = (w >> 22u) ^ w;
    return (float)state / 4294967296.0f;
}

// comp:402-417
VRT_DEV F3 cosine_hemisphere(F3 n, float rx, float ry) {
    const float kPI = 3.14159265359f;
    float phi = 2.0f * kPI * ry;
    float ct = __builtin_sqrtf(rx);
    float stheta = __builtin_sqrtf(1.0f - rx);
    float sn, cs;
    det_sincos(phi, sn, cs);
    float x = stheta * cs;
    float z = stheta * sn;
    F3 up = __builtin_fabsf(n.z) < 0.999f ? F3{0.0f, 0.0f, 1.0f} : F3{1.0f, 0.0f, 0.0f};
    F3 tangent = normalize3(cross3(up, n));
    F3 bitangent = cross3(n, tangent);
    F3 r = add3(add3(scale3(tangent, x), scale3(bitangent, z)), scale3(n, ct));
    return normalize3(r);
}

// GLSL refract / reflect
VRT_DEV F3 refract3(F3 I, F3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return F3{0.0f, 0.0f, 0.0f};
    float f = eta * d + __builtin_sqrtf(k);
    return sub3(scale3(I, eta), scale3(N, f));
}
VRT_DEV F3 reflect3(F3 I, F3 N) { float d = dot3(N, I); return sub3(I, scale3(N, 2.0f * d)); }

VRT_DEV void absorb(float tc[3], float density, float dist, const float mc[3]) {  // comp:482-486,512-516
    float k = -density * dist;
#pragma unroll
    for (int i = 0; i < 3; ++i) tc[i] = tc[i] * det_expf(k * (1.0f - mc[i]));
}

// byte form of a ray's IOF for the byte-based medium test (vrt_kernels.hip.h): every IOF this shader
// produces is r(b) = (b/255)*3 for a byte b, or 1.0 = r(85); r(b)*85 is within 1e-5 of b
VRT_DEV uint32_t iof_to_byte(float iof) { return (uint32_t)__builtin_rintf(iof * 85.0f); }

VRT_DEV RayS make_ray(F3 o, F3 d, float iof, float w, const float tint[3], float dim, const float mc[3], float md, int depth) {
    RayS r;
    r.o = o; r.d = d; r.iof = iof; r.weight = w;
    r.tint[0] = tint[0]; r.tint[1] = tint[1]; r.tint[2] = tint[2];
    r.dim = dim;
    r.mc[0] = mc[0]; r.mc[1] = mc[1]; r.mc[2] = mc[2];
    r.md = md; r.depth = depth;
    return r;
}

// One record of the deferred-bounce queues (KArgs::defer_rec): the lanes that are here together take consecutive slots
// of queue `queue` with one atomic between them (ballot + popcount: the wave's aggregate), then each writes its planes.
VRT_DEV void defer_bounce(const KArgs &a, uint32_t queue, uint32_t out_offset, F3 o, F3 d, const float tint[3], const float fc[3], float iof,
                          float weight, const float mc[3], float md) {
    const uint64_t here = __builtin_amdgcn_ballot_w64(true);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
    uint32_t base = 0u;
    if (rank == 0u) base = atomicAdd(&a.defer_count[queue * kDeferStride], (uint32_t)__builtin_popcountll(here));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);   // the first active lane is the one of rank 0
    const size_t plane = (size_t)kDeferQueues * a.defer_cap;
    float *r = a.defer_rec + (size_t)queue * a.defer_cap + base + rank;
    r[0 * plane] = o.x; r[1 * plane] = o.y; r[2 * plane] = o.z;
    r[3 * plane] = d.x; r[4 * plane] = d.y; r[5 * plane] = d.z;
    r[6 * plane] = tint[0]; r[7 * plane] = tint[1]; r[8 * plane] = tint[2];
    r[9 * plane] = fc[0]; r[10 * plane] = fc[1]; r[11 * plane] = fc[2];
    r[12 * plane] = iof; r[13 * plane] = weight;
    r[14 * plane] = mc[0]; r[15 * plane] = mc[1]; r[16 * plane] = mc[2];
    r[17 * plane] = md;
    r[18 * plane] = __uint_as_float(out_offset);
}

// DEFER: when the ray stack is empty at the moment a diffuse bounce would be pushed, that bounce is the last ray of the
// pixel -- nothing is accumulated after it -- so it goes to a queue instead, together with the colour summed so far;
// bounce_kernel (vrt_bounce.hip.h) marches it among full waves of such rays and writes the pixel. A bounce ray (depth 1)
// spawns nothing (comp:590-594), so the queue is one level deep. The accumulation order per pixel is unchanged.
template <class TRAV, bool DEFER>
__device__ void trace_pixel_full(const KArgs &a, const View &vw, const typename TRAV::Ctx &tc_, int px, int py, uint32_t &rgba, int2 &idd, LateOut &lo,
                                 uint32_t queue, uint32_t out_offset) {
    const float kPI = 3.14159265359f;
    const float sky[3] = {0.5f, 0.7f, 1.0f};
    const float kSun = 3.0f;
    uint32_t rng = rng_init(px, py, 0);
    const F3 ray_dir = primary_ray_dir(a, vw, px, py);
    const F3 ray_origin{vw.cam_pos[0], vw.cam_pos[1], vw.cam_pos[2]};

    int voxel_id = 0;
    int pixel_dist = a.wmax[0] - a.wmin[0];
    F3 gro = scale3(ray_origin, a.voxel_scale);
    Decoded tv = decode_leaf(vw.eye0, vw.eye1);  // medium at the eye: looked up once by the dispatcher
    float start_iof = (tv.p[0] > 0.0f && tv.p[0] < 3.0f) ? tv.p[0] : 1.0f;

    // The ray stack (comp:451) lives in private memory: 8 x 68 bytes per lane. (Holding the entry pushed last in registers
    // until it is popped -- no scratch traffic at all for opaque scenes -- was measured: the 17 extra live values push
    // the 96-register build into 20 spills and the frame from 0.261 to 0.282 ms.)
    StackRay stack[kMaxRays];
    // the primary ray starts in registers, and so does a diffuse bounce pushed on an empty stack (every bounce of an opaque
    // scene): pushed and popped at once they made a store -> load round trip through scratch, 17 words each way per lane
    RayS r;
    {
        const float ones[3] = {1.0f, 1.0f, 1.0f};
        const float gl3[3] = {a.global_light[0], a.global_light[1], a.global_light[2]};
        r = make_ray(gro, ray_dir, start_iof, 1.0f, gl3, 0.0f, tv.c[3] > 0.0f ? tv.c : ones, tv.c[3] * 5.0f, 0);
    }
    int sp = 0;
    bool in_regs = true;   // `r` already holds the ray to march next: the primary ray, or a diffuse bounce pushed on an empty stack
    const auto push = [&](const RayS &nr, uint32_t medium) {
        StackRay e;
        e.o = nr.o; e.d = nr.d; e.iof = nr.iof; e.weight = nr.weight;
        e.tint[0] = nr.tint[0]; e.tint[1] = nr.tint[1]; e.tint[2] = nr.tint[2];
        e.dim = nr.dim; e.medium = medium;
        stack[sp++] = e;
    };
    const auto pop = [&]() {
        const StackRay e = stack[--sp];
        r.o = e.o; r.d = e.d; r.iof = e.iof; r.weight = e.weight;
        r.tint[0] = e.tint[0]; r.tint[1] = e.tint[1]; r.tint[2] = e.tint[2];
        r.dim = e.dim;
        r.mc[0] = unorm_of((float)(e.medium & 0xffu)); r.mc[1] = unorm_of((float)((e.medium >> 8) & 0xffu));
        r.mc[2] = unorm_of((float)((e.medium >> 16) & 0xffu));
        r.md = unorm_of((float)(e.medium >> 24)) * 5.0f;
        r.depth = 0;
    };
    bool deferred = false;
    float fc[3] = {0.0f, 0.0f, 0.0f};
    const float *gl = a.global_light;
    const F3 light{a.light_dir[0], a.light_dir[1], a.light_dir[2]};
    // x / PI in its in-range form where the dispatcher vouches for the lights' range (KArgs::shade_fast, div_pi_inrange())
    const bool shade_fast = a.shade_fast != 0;
    const auto over_pi = [&](float x) { return shade_fast ? div_pi_inrange(x) : x / kPI; };

#ifdef VRT_EXP_STATS   // experiment builds only (tools/room_stats.sh): what the wave's time is made of, left in tile_cost
    unsigned long long st_acc = 0;
    const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
    while (in_regs || sp > 0) {
#ifdef VRT_EXP_STATS
        const unsigned long long st_tp = __builtin_readcyclecounter();
#endif
        if (!in_regs) pop();
        in_regs = false;
        Hit h;
#ifdef VRT_EXP_STATS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_t0 = __builtin_readcyclecounter();
        if (VRT_EXP_STATS == 10) st_acc += st_t0 - st_tp;
        if (VRT_EXP_STATS == 1) st_acc += 1;
#endif
        const bool hit = TRAV::march(a, tc_, r.o, r.d, r.iof, iof_to_byte(r.iof), h);
#ifdef VRT_EXP_STATS
        if (VRT_EXP_STATS == 2) st_acc += __builtin_readcyclecounter() - st_t0;
        if (VRT_EXP_STATS == 5) {   // the wave's trips: the longest march of the round
            int m = h.iters;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(m, off); m = o > m ? o : m; }
            st_acc += (unsigned long long)m;
        }
        if (VRT_EXP_STATS == 6) st_acc += (unsigned long long)h.iters;   // this lane's own trips
        const unsigned long long st_t2 = __builtin_readcyclecounter();
        unsigned long long st_t3 = st_t2;
        // 7: from the end of march() to the translucent / opaque decision (misses end here); 8: the translucent branch; 9: the opaque one
        // without its shadow ray. Lanes that `continue` out of a segment do not count it: the maximum over lanes is a lane that stayed.
#define VRT_ST_MARK(k) do { const unsigned long long st_now = __builtin_readcyclecounter(); if (VRT_EXP_STATS == (k)) st_acc += st_now - st_t3; st_t3 = st_now; } while (0)
#else
#define VRT_ST_MARK(k) do { } while (0)
#endif
        float tc[3] = {r.tint[0], r.tint[1], r.tint[2]};
        if (!hit && r.depth <= 0) {
            if (r.dim > 1e-6f && r.md > 0.0f) absorb(tc, r.md, r.dim, r.mc);
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + gl[k] * sky[k] * tc[k] * r.weight;
            continue;
        } else if (!hit) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(tc[k] * sky[k] * kSun * r.weight);
            continue;
        }
        const F3 hn{h.axis == 0 ? h.n : 0.0f, h.axis == 1 ? h.n : 0.0f, h.axis == 2 ? h.n : 0.0f};
        F3 normal = hn;
        if (!(len3(hn) > 0.0f)) normal = F3{0.0f, 1.0f, 0.0f};
        const F3 hp = h.point;
        // x / 1.0f == x: at the reference's u_voxelScale of 1 the four divisions are skipped (a wave-uniform branch)
        F3 hpw = hp;
        if (a.voxel_scale != 1.0f) {
            hpw = F3{hp.x / a.voxel_scale, hp.y / a.voxel_scale, hp.z / a.voxel_scale};
            r.dim = r.dim + len3(sub3(hpw, r.o)) / a.voxel_scale;
        } else {
            r.dim = r.dim + len3(sub3(hpw, r.o));
        }
        Decoded hv = decode_leaf(h.h0, h.h1);
        Decoded last = decode_leaf(h.p0, h.p1);
        if (hv.c[3] <= 0.0f) { hv.p[0] = 1.0f; hv.p[1] = 0.0f; hv.p[2] = 0.0f; }
        if (last.c[3] <= 0.0f) {
            if (r.iof > 0.0f) { last.p[0] = 0.0f; last.p[1] = 0.0f; last.p[2] = 0.0f; }
            else { last.p[0] = 1.0f; last.p[1] = 0.0f; last.p[2] = 0.0f; }
        }
        float sc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sc[k] = hv.c[3] > 0.0f ? hv.c[k] : last.c[k];
        float n2 = hv.p[0] > 0.0f ? hv.p[0] : 1.0f;
        float n1 = last.p[0] > 0.0f ? last.p[0] : 1.0f;
        const F3 inc = r.d;
        if (r.dim > 1e-6f && r.md > 0.0f) absorb(tc, r.md, r.dim, r.mc);
        if (h.map.x == a.highlighted[0] && h.map.y == a.highlighted[1] && h.map.z == a.highlighted[2]) {
            sc[0] = 1.0f - sc[0]; sc[1] = 1.0f - sc[1]; sc[2] = 1.0f - sc[2]; sc[3] = 1.0f;
        }
        const float cosi = dot3(inc, normal);
        if (cosi > 0.0f) { normal = F3{-normal.x, -normal.y, -normal.z}; const float t = n1; n1 = n2; n2 = t; }
        const float ndotl = fmax_c(dot3(normal, light), 0.0f);

        if (r.depth == 0 && voxel_id == 0 && sc[3] >= 1.0f) {  // comp:539-544
            const int lin = h.map.x + a.tex_dim * (h.map.y + a.tex_dim * h.map.z);
            voxel_id = lin * 6 + face_index(h.axis, h.n);
            pixel_dist = (int)len3(sub3(hpw, ray_origin));
        }

        VRT_ST_MARK(7);
        if (r.depth <= 0 && sc[3] < 1.0f) {  // translucent, comp:547-572
            const F3 refr_dir = refract3(inc, normal, n1 / n2);
            const float R0 = (n1 - n2) / (n1 + n2) * (n1 - n2) / (n1 + n2);
            const F3 ninc{-inc.x, -inc.y, -inc.z};
            const float cos_t = fmax_c(0.0f, dot3(ninc, normal));
            float fres = R0 + (1.0f - R0) * det_powf(1.0f - cos_t, 5.0f);
            fres = fmin_c(fmax_c(fres, 0.0f), 1.0f);
            const bool has_tir = len3(refr_dir) < 0.001f;
            const float reflect_i = fres;
            const float refract_i = has_tir ? 0.0f : (1.0f - fres);
            if (sp == kMaxRays || reflect_i <= 0.001f || refract_i <= 0.001f) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float direct = gl[k] * ndotl;
                    const float lit = sc[k] * direct;
                    fc[k] = fc[k] + tc[k] * lit * r.weight;
                }
                continue;
            }
            if (reflect_i > 0.001f && sp < kMaxRays) {
                const float rw = r.weight * reflect_i;
                if (rw > 1e-4f)
                    push(make_ray(add3(hp, scale3(normal, 1e-4f)), reflect3(inc, normal), n1, rw, tc, r.dim, last.c, last.c[3] * 5.0f, r.depth), h.p0);
            }
            if (refract_i > 0.001f && sp < kMaxRays && !has_tir) {
                // the last push of this iteration, hence the next ray marched: it takes r's registers (see `in_regs`)
                r = make_ray(sub3(hp, scale3(normal, 1e-4f)), refr_dir, n2, r.weight * refract_i, tc, 0.0f, hv.c, hv.c[3] * 5.0f, r.depth);
                in_regs = true;
            }
            VRT_ST_MARK(8);
        } else {  // opaque, comp:573-618
            const float emission = hv.p[1] * 10.0f;
            if (emission > 0.0f && r.depth == 0) {
#pragma unroll
                for (int k = 0; k < 3; ++k) fc[k] = fc[k] + tc[k] * sc[k] * emission * r.weight;
                continue;
            } else if (emission > 0.0f) {
#pragma unroll
                for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(tc[k] * sc[k] * emission * r.weight);
                continue;
            }
            if (r.depth == 0) {
                // (the dispatcher's LightSetup, which the primary + shadow kernel takes, costs this one 135 instructions per wave:
                // at its register budget the uniform values are re-materialised inside the shadow loop)
                VRT_ST_MARK(9);
#ifdef VRT_EXP_STATS
                const unsigned long long st_t1 = __builtin_readcyclecounter();
#endif
                const int lit = TRAV::shadow(a, tc_, add3(hp, scale3(normal, 2e-3f)), light, h);
#ifdef VRT_EXP_STATS
                if (VRT_EXP_STATS == 3) st_acc += __builtin_readcyclecounter() - st_t1;
                st_t3 = __builtin_readcyclecounter();
#endif
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float direct = gl[k] * (float)lit * ndotl;
                    fc[k] = fc[k] + over_pi(direct * sc[k] * tc[k] * r.weight);
                }
            } else {
                const float amb = fmax_c(1.0f - det_expf(-r.dim / 512.0f), 0.01f);
#pragma unroll
                for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(amb * sc[k] * tc[k] * r.weight);
                continue;
            }
            for (int i = 0; i < kIndirectSamples && sp < kMaxRays && r.depth <= kBounces; ++i) {
                const float rx = rng_next(rng), ry = rng_next(rng);
                const F3 bd = cosine_hemisphere(normal, rx, ry);
                const float nw = r.weight / (float)kIndirectSamples;
                const float tint[3] = {tc[0] * sc[0], tc[1] * sc[1], tc[2] * sc[2]};
                static_assert(kIndirectSamples == 1, "the bounce ray takes the finished ray's registers: one per hit");
                if (DEFER && sp == 0) {
                    defer_bounce(a, queue, out_offset, add3(hp, scale3(normal, 1e-1f)), bd, tint, fc, n1, nw, last.c, last.c[3] * 5.0f);
                    deferred = true;
                } else {
                    // Pushed last, it is the ray popped next -- whatever else waits on the stack (the reflections of the glass
                    // the primary ray came through): it takes r's registers instead of a round trip through scratch, 17 words each
                    // way per lane with the march waiting on the reload. (Round 2 did this for an empty stack only.)
                    r = make_ray(add3(hp, scale3(normal, 1e-1f)), bd, n1, nw, tint, 0.0f, last.c, last.c[3] * 5.0f, r.depth + 1);
                    in_regs = true;
                }
            }
            VRT_ST_MARK(9);
        }
    }
#ifdef VRT_EXP_STATS
    if (VRT_EXP_STATS == 4) st_acc = __builtin_readcyclecounter() - st_begin;
    {
        uint32_t m = (uint32_t)st_acc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off); m = o > m ? o : m; }
        if (a.tile_cost && (px & 7) == 0 && (py & 7) == 0) a.tile_cost[(py >> 3) * ((a.width + 7) >> 3) + (px >> 3)] = m;
    }
#endif
    rgba = unorm8(fc[0]) | (unorm8(fc[1]) << 8) | (unorm8(fc[2]) << 16) | (255u << 24);
    idd = make_int2(voxel_id, pixel_dist);
    lo = late_out(late_args(), late_view());
    lo.skip_rgba = deferred;
}

// ---- the two-pass form for opaque scenes ------------------------------------------------------------------------------------
// In a scene without translucent voxels seen from empty space, pathTrace (comp:435-622) is: the primary ray; on an opaque,
// non-emissive hit a shadow ray and ONE diffuse bounce ray, which spawns nothing (comp:590-594). Pass 1 (trace_kernel MODE 4) is the
// primary + shadow kernel -- 64 VGPRs, seven waves per SIMD, no ray stack -- leaving a 20-byte seed per pixel; pass 2 (MODE 5,
// this function) marches the bounce rays of the seeded pixels in the same 8 x 8 tiles (the coherence the one-kernel form has),
// again without a stack, and writes the pixel's final colour. The accumulation order per pixel is pathTrace's: direct term, then
// the bounce's term. Everything pass 2 recomputes is computed from the same inputs by the same operations as pass 1 did.
template <class TRAV>
__device__ bool bounce_pixel(const KArgs &a, const typename TRAV::Ctx &tc_, int px, int py, const Seed seed, uint32_t &rgba) {
    const uint32_t word = seed.word;
    const bool valid = (word & kSeedValid) != 0u;
    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return false;   // a tile without bounces (sky, emissive surfaces)
    if (!valid) return false;
    const float kPI = 3.14159265359f;
    const float sky[3] = {0.5f, 0.7f, 1.0f};
    const float kSun = 3.0f;
    const F3 hp = seed.hp;
    const float n1 = seed.iof;
    const bool shade_fast = a.shade_fast != 0;
    const auto over_pi = [&](float x) { return shade_fast ? div_pi_inrange(x) : x / kPI; };
    // the surface colour and the normal as pass 1 used them
    float sc[3] = {unorm_of((float)(word & 0xffu)), unorm_of((float)((word >> 8) & 0xffu)), unorm_of((float)((word >> 16) & 0xffu))};
    if (word & (1u << 27)) { sc[0] = 1.0f - sc[0]; sc[1] = 1.0f - sc[1]; sc[2] = 1.0f - sc[2]; }
    const int naxis = (int)((word >> 24) & 3u);
    const float nval = (word & (1u << 26)) ? -1.0f : 1.0f;
    const F3 normal{naxis == 0 ? nval : 0.0f, naxis == 1 ? nval : 0.0f, naxis == 2 ? nval : 0.0f};
    const float *gl = a.global_light;
    const F3 light{a.light_dir[0], a.light_dir[1], a.light_dir[2]};
    const float ndotl = fmax_c(nval * comp(light, naxis), 0.0f);
    const float lit = (word & (1u << 28)) ? 1.0f : 0.0f;
    float fc[3], tint[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // comp:587-589 as trace_pixel<1> evaluates it (the throughput of a primary ray from empty space is the light)
        const float direct = gl[k] * lit * ndotl;
        fc[k] = 0.0f + over_pi(direct * sc[k] * gl[k] * 1.0f);
        tint[k] = gl[k] * sc[k];
    }
    // comp:596-616: the first two random numbers of the pixel
    uint32_t rng = rng_init(px, py, 0);
    const float rx = rng_next(rng), ry = rng_next(rng);
    const F3 bd = cosine_hemisphere(normal, rx, ry);
    const F3 ro = add3(hp, scale3(normal, 1e-1f));
    Hit h;
    // A bounce ray starts in the medium in front of the surface: empty space (n1 = 1.0), which the one-loop march of views from
    // empty space (TRAV::Eye85) takes -- except behind a flipped normal (an axis-aligned ray on a zero direction component, comp:497,
    // 524: n1 is then the hit voxel's index). A wave that holds such a lane marches all its rays with the general loop.
    bool hit;
    if (__builtin_amdgcn_ballot_w64(n1 != 1.0f) == 0ull) hit = TRAV::Eye85::march(a, tc_, ro, bd, 1.0f, 85u, h);
    else hit = TRAV::General::march(a, tc_, ro, bd, n1, iof_to_byte(n1), h);
    if (!hit) {   // depth 1 (comp:489-495)
#pragma unroll
        for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(tint[k] * sky[k] * kSun * 1.0f);
    } else {
        F3 hpw = h.point;
        float dim;
        if (a.voxel_scale != 1.0f) {
            hpw = F3{h.point.x / a.voxel_scale, h.point.y / a.voxel_scale, h.point.z / a.voxel_scale};
            dim = 0.0f + len3(sub3(hpw, ro)) / a.voxel_scale;
        } else {
            dim = 0.0f + len3(sub3(hpw, ro));
        }
        Decoded hv = decode_leaf(h.h0, h.h1);
        const Decoded last = decode_leaf(h.p0, h.p1);
        if (hv.c[3] <= 0.0f) { hv.p[1] = 0.0f; }
        float s2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) s2[k] = hv.c[3] > 0.0f ? hv.c[k] : last.c[k];
        if (h.map.x == a.highlighted[0] && h.map.y == a.highlighted[1] && h.map.z == a.highlighted[2]) {
            s2[0] = 1.0f - s2[0]; s2[1] = 1.0f - s2[1]; s2[2] = 1.0f - s2[2];
        }
        const float emission = hv.p[1] * 10.0f;
        if (emission > 0.0f) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(tint[k] * s2[k] * emission * 1.0f);
        } else {
            const float amb = fmax_c(1.0f - det_expf(-dim / 512.0f), 0.01f);
#pragma unroll
            for (int k = 0; k < 3; ++k) fc[k] = fc[k] + over_pi(amb * s2[k] * tint[k] * 1.0f);
        }
    }
    rgba = unorm8(fc[0]) | (unorm8(fc[1]) << 8) | (unorm8(fc[2]) << 16) | (255u << 24);
    return true;
}

}  // namespace full
}  // namespace vrt
