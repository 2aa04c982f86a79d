// vrt_common.hip.h -- device-side pieces shared by every traversal variant:
// kernel arguments, the fp32 arithmetic conventions, ray generation and shading
// of shaders/raytracing.comp (main :624-645, the primary-ray subset of pathTrace
// :435-497, 522-544, 573-589, 619-621), and the pixel-tile kernel skeleton.
//
// Arithmetic contract (must hold bit-for-bit against the CPU oracle):
//   * IEEE binary32, round-to-nearest-even, NO contraction (-ffp-contract=off),
//     correctly rounded '/' and sqrtf (hipcc default), denormals kept.
//   * mat4*vec4 = (m0*x + m1*y) + (m2*z + m3*w); dot3 = (x*x' + y*y') + z*z';
//     normalize(v) = v * (1/sqrt(dot(v,v))); min(a,b) = b<a?b:a; max(a,b) = a<b?b:a.
//   * The DDA advances one octree NODE per step exactly as hitMarching does; the
//     node lookup may use any structure because octreeFind's result is a pure
//     function of the query point (the deepest node containing it).
#pragma once

#include "vrt_args.h"

namespace vrt {

#define VRT_DEV __device__ __forceinline__

VRT_DEV float fmin_c(float a, float b) { return b < a ? b : a; }
VRT_DEV float fmax_c(float a, float b) { return a < b ? b : a; }
VRT_DEV float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VRT_DEV float len3(F3 a) { return __builtin_sqrtf(dot3(a, a)); }
VRT_DEV F3 scale3(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
VRT_DEV F3 add3(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VRT_DEV F3 sub3(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VRT_DEV F3 normalize3(F3 a) { return scale3(a, 1.0f / __builtin_sqrtf(dot3(a, a))); }

// 1/x and sqrt(x) for arguments known to be in range: the instruction sequences the compiler emits for the correctly
// rounded `1.0f / x` and sqrtf(x), WITHOUT their range handling -- the two v_div_scale, v_div_fmas' scale and v_div_fixup;
// the 2^32 pre-scale, its undo and the zero/infinity class test -- 7 and 9 instructions instead of 12 and 17, and the
// ones that go are the dear kinds (profiles/r02_valu_rate.txt). Inside the stated ranges those steps are identities, so
// the results are the same bits (checked against correctly rounded values: vrt_test_math ops 30 and 31, csrc/test/vrt_test.hip).
//   rcp_inrange : 2^-95 <= |x| < 2^126   (v_div_scale_f32 leaves 1.0 and x alone there)
//   sqrt_inrange: 2^-96 <= x < infinity
VRT_DEV float rcp_inrange(float x) {
    float y = __builtin_amdgcn_rcpf(x);
    y = __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
    const float q = __builtin_fmaf(__builtin_fmaf(-x, y, 1.0f), y, y);
    return __builtin_fmaf(__builtin_fmaf(-x, q, 1.0f), y, q);
}
VRT_DEV float sqrt_inrange(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r = (0.0f >= __builtin_fmaf(-below, s, x)) ? below : s;
    return (0.0f < __builtin_fmaf(-above, s, x)) ? above : r;
}
// x / 3.14159265359f (comp:589, the Lambert term) for 0 <= x < 2^97: the same sequence with its denominator known when
// the code is compiled -- five fused operations. y is the refined reciprocal the sequence would compute from v_rcp_f32;
// the sequence's result is the correctly rounded quotient for any starting value within an ulp of 1/d, so the correctly
// rounded 1/PI serves as one. Below 2^-103 the hardware form rescales and this one may differ in the last bit: such an
// x is a colour term that rounds to 0 in rgba8 either way. (vrt_test_math op 32: all 2^23 mantissas and random exponents.)
VRT_DEV float div_pi_inrange(float x) {
    constexpr float kD = 3.14159265359f;
    constexpr float kY0 = 0.318309873342514038f;   // 0x3ea2f983 = RN(1 / kD)
    const float y = __builtin_fmaf(__builtin_fmaf(-kD, kY0, 1.0f), kY0, kY0);
    float q = x * y;
    q = __builtin_fmaf(__builtin_fmaf(-kD, q, x), y, q);
    return __builtin_fmaf(__builtin_fmaf(-kD, q, x), y, q);
}
VRT_DEV F3 normalize3_inrange(F3 a) { return scale3(a, rcp_inrange(sqrt_inrange(dot3(a, a)))); }
VRT_DEV float sign_c(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
VRT_DEV float comp(F3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
VRT_DEV I3 floor_i3(F3 p) { return I3{(int)__builtin_floorf(p.x), (int)__builtin_floorf(p.y), (int)__builtin_floorf(p.z)}; }

VRT_DEV bool in_world(const KArgs &a, I3 p) {  // comp:224-226
    return p.x >= a.wmin[0] && p.y >= a.wmin[1] && p.z >= a.wmin[2] &&
           p.x < a.wmax[0] && p.y < a.wmax[1] && p.z < a.wmax[2];
}

// Leaf record words (vrt_layout.h): w0 = R | G<<8 | B<<16 | alpha<<24, w1 = refr | illum<<8 | k<<16;
// 0/0 stands for empty space (the shader's zeroed VoxelData).
VRT_DEV float refraction_of(uint32_t w1) { return ((float)(w1 & 0xffu) / 255.0f) * 3.0f; }  // comp:126-128,177

struct Hit {
    I3 map;           // hitMapPos
    F3 point;         // hitPoint
    int axis;         // hitNormal = n on this axis, 0 elsewhere (comp:292-294)
    float n;          // -sign(rayDir[axis]): +1, -1, or -0 for a zero direction component
    uint32_t p0, p1;  // prevVoxel leaf words
    uint32_t h0, h1;  // hitVoxel leaf words
    // where the traversal stood when the march ended (meaning private to the TRAV policy): lets the shadow
    // ray, which starts 2e-3 off the hit point, begin its first lookup there instead of at the root
    uint32_t r_node, r_anode;
    int r_s, r_as;
    I3 r_last;
#ifdef VRT_EXP_STATS
    int iters;        // experiment builds: the march loop's trip count of this lane
#endif
};

// The shadow ray's direction-dependent constants (comp:335-345), uniform over a launch: see KArgs::light_inv.
struct LightSetup { F3 dir, inv, push, dposf; I3 dpos; };

// exp() convention shared with the oracle (Cephes-style, plain mul/add)
VRT_DEV float det_expf(float x) {
    if (x > 88.0f) return __builtin_huge_valf();
    if (x < -87.0f) return 0.0f;
    float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = x - k * 0.693359375f;
    r = r - k * -2.12194440e-4f;
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * z + r;
    p = p + 1.0f;
    int ki = (int)k;
    return p * __uint_as_float((uint32_t)(ki + 127) << 23);
}

// getFaceIndex (comp:419-433) of an axis normal: length(normal) is |n| (1 or 0), the dominant axis is `axis`
VRT_DEV int face_index(int axis, float n) {
    if (!(__builtin_fabsf(n) >= 0.5f)) return 0;
    return axis * 2 + (n > 0.0f ? 0 : 1);
}

VRT_DEV uint32_t unorm8(float v) {  // rgba8 imageStore: clamp, scale, round to nearest even
    float c = fmin_c(fmax_c(v, 0.0f), 1.0f);
    return (uint32_t)__builtin_rintf(c * 255.0f);
}

VRT_DEV void mat_vec(const float *m, float x, float y, float z, float w, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = (m[0 * 4 + r] * x + m[1 * 4 + r] * y) + (m[2 * 4 + r] * z + m[3 * 4 + r] * w);
}

// (float)b / 255.0f for a byte b, bit for bit, without the division: the product with the rounded reciprocal is off
// by one ulp for 126 of the 256 bytes, and one residual step (both fused, as written) repairs all of them -- checked
// exhaustively against the correctly rounded quotient (tests/test_oracle.py) and through every parity frame.
VRT_DEV float unorm_of(float b) {
    constexpr float kRcp255 = 0.003921568859368563f;  // 0x3b808081 = RN(1/255)
    const float q = b * kRcp255;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, b), kRcp255, q);
}

// Leaf words -> the shader's VoxelData floats (comp:173-178).
struct Decoded { float c[4]; float p[3]; };
VRT_DEV Decoded decode_leaf(uint32_t w0, uint32_t w1) {
    Decoded d;
    d.c[0] = unorm_of((float)(w0 & 0xffu));
    d.c[1] = unorm_of((float)((w0 >> 8) & 0xffu));
    d.c[2] = unorm_of((float)((w0 >> 16) & 0xffu));
    d.c[3] = unorm_of((float)(w0 >> 24));
    d.p[0] = unorm_of((float)(w1 & 0xffu)) * 3.0f;
    d.p[1] = unorm_of((float)((w1 >> 8) & 0xffu));
    d.p[2] = unorm_of((float)((w1 >> 16) & 0xffu));
    return d;
}

// The kernel's own arguments, re-read from the kernarg segment (KArgs is trace_kernel's first parameter, so it sits at
// offset 0). Uniforms that are needed only after the march -- lights, the highlighted voxel, tex_dim -- are fetched
// through this at shading time instead of living in scalar registers across the traversal loops; the empty asm
// keeps the compiler from hoisting the loads back above its position.
typedef const KArgs __attribute__((address_space(4))) *LateArgs;
VRT_DEV LateArgs late_args() {
    LateArgs p = (LateArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}

// The launch's views the same way: ViewSet is the second parameter and follows KArgs at its natural alignment.
typedef const View __attribute__((address_space(4))) *LateView;
VRT_DEV LateView late_view() {
    constexpr size_t kOffset = (sizeof(KArgs) + alignof(ViewSet) - 1) / alignof(ViewSet) * alignof(ViewSet);
    const char __attribute__((address_space(4))) *p =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + kOffset + blockIdx.y * sizeof(View);
    asm volatile("" : "+s"(p));
    return (LateView)p;
}

// Where a pixel's two results go: read together with the other late arguments (one scalar-load round trip).
struct LateOut {
    uint32_t *out_rgba;
    int2 *out_id;
    int width, compact;
    bool skip_rgba;   // MODE 3: the pixel's colour is finished by bounce_kernel
};
VRT_DEV LateOut late_out(LateArgs la, LateView lv) { return LateOut{lv->out_rgba, lv->out_id, la->width, la->compact, false}; }

// Ray generation (comp:624-641) and pathTrace's own normalisation of the direction (comp:441). Two forms, chosen per view
// by the dispatcher (wave-uniform): the shader's operations one by one, or -- View::gen_fast -- the same operations with
// everything that depends on the column or the row alone read from the view's tables (the two index divisions, the
// first matrix product and the perspective divide: 8 of the prologue's 15 divisions) and the remaining 1/x and sqrt in
// their in-range forms. Same bits either way (tests: frames of both forms against the oracle).
VRT_DEV F3 primary_ray_dir(const KArgs &a, const View &vw, int px, int py) {
    if (vw.gen_fast) {
        const F3 vd = normalize3_inrange(F3{vw.gen_x[px], vw.gen_y[py], vw.gen_z});
        float wd4[4];
        mat_vec(vw.inv_view, vd.x, vd.y, vd.z, 0.0f, wd4);
        const F3 d = normalize3_inrange(F3{wd4[0], wd4[1], wd4[2]});
        return scale3(d, rcp_inrange(sqrt_inrange(dot3(d, d))));
    }
    float u = ((float)px / (float)a.width) * 2.0f - 1.0f;
    float v = ((float)py / (float)a.height) * 2.0f - 1.0f;
    float view[4];
    mat_vec(vw.inv_proj, u, v, -1.0f, 1.0f, view);
    if (__builtin_fabsf(view[3]) > 1e-6f) { float w = view[3]; view[0] = view[0] / w; view[1] = view[1] / w; view[2] = view[2] / w; view[3] = view[3] / w; }
    F3 vd = normalize3(F3{view[0], view[1], view[2]});
    float wd4[4];
    mat_vec(vw.inv_view, vd.x, vd.y, vd.z, 0.0f, wd4);
    const F3 d = normalize3(F3{wd4[0], wd4[1], wd4[2]});
    return scale3(d, 1.0f / __builtin_sqrtf(dot3(d, d)));
}

// traversals whose shadow() takes the dispatcher's LightSetup declare `static constexpr bool kHostLight = true`
template <class T, class = void> struct host_light { static constexpr bool value = false; };
template <class T> struct host_light<T, decltype((void)T::kHostLight)> { static constexpr bool value = T::kHostLight; };

// The seed pass 1 of the two-pass full path tracer (trace_kernel MODE 4) leaves per pixel for pass 2 (MODE 5, vrt_full.hip.h
// bounce_pixel): five words, tile-major planes of 64 lanes -- seed[(tile * 5 + plane) * 64 + lane] -- so a wave's store of a plane is
// one 256-byte line. Planes 0-2: the hit point (floats); plane 4: the bounce ray's index of refraction (float); plane 3:
//   [23:0] the surface colour's three bytes (the hit voxel's, or the previous voxel's where comp:505-507 takes that one)
//   [25:24] axis and [26] sign (1 = negative) of the shading normal after the flip of comp:524  [27] colour inverted (highlighted voxel)
//   [28] the shadow ray's answer (lit)  [31] valid: this pixel has a diffuse bounce to march (an opaque, non-emissive first hit)
constexpr uint32_t kSeedPlanes = 5, kSeedValid = 1u << 31;
struct Seed { F3 hp; uint32_t word; float iof; };   // the same in registers (MODE 6: both passes in one wave, nothing through memory)

// One pixel: ray generation (comp:624-641), primary-ray pathTrace, packing of the two outputs.
// TRAV supplies the traversal: march(), shadow(). MODE: 0 primary, 1 primary + shadow ray. seed (MODE 1 only): see above.
template <int MODE, class TRAV>
VRT_DEV void trace_pixel(const KArgs &a, const View &vw, const typename TRAV::Ctx &tc_, int px, int py, uint32_t &rgba, int2 &idd, LateOut &lo,
                         uint32_t *seed = nullptr, Seed *seed_regs = nullptr) {
    const float kPI = 3.14159265359f;
    const F3 ray_dir = primary_ray_dir(a, vw, px, py);
    F3 ray_origin{vw.cam_pos[0], vw.cam_pos[1], vw.cam_pos[2]};

    int voxel_id = 0;
    int pixel_dist = a.wmax[0] - a.wmin[0];
    F3 gro = scale3(ray_origin, a.voxel_scale);
    // medium at the eye (comp:445-449)
    // medium at the eye (comp:445-449): the same node for every ray of the view, found once by the dispatcher
    Decoded tvd = decode_leaf(vw.eye0, vw.eye1);
    float start_iof = (tvd.p[0] > 0.0f && tvd.p[0] < 3.0f) ? tvd.p[0] : 1.0f;
    float medium_density = tvd.c[3] * 5.0f;
    float mc[3] = {1.0f, 1.0f, 1.0f};
    if (tvd.c[3] > 0.0f) { mc[0] = tvd.c[0]; mc[1] = tvd.c[1]; mc[2] = tvd.c[2]; }
    float fc[3] = {0.0f, 0.0f, 0.0f};
    const float sky[3] = {0.5f, 0.7f, 1.0f};

    Hit h;
    uint32_t seed_word = 0u;
    float seed_iof = 1.0f;
    // byte form of start_iof for traversals that test media on bytes: r(b) in (0, 3) <=> 1 <= b <= 254, else 1.0 == r(85)
    const uint32_t eye_b = vw.eye1 & 0xffu;
    const uint32_t iof_byte = (eye_b >= 1u && eye_b <= 254u) ? eye_b : 85u;
    bool hit = TRAV::march(a, tc_, gro, ray_dir, start_iof, iof_byte, h, &vw);
    // The uniforms of the shading stage. With a shadow march still to come they are re-read from the kernarg segment
    // here, in one scalar-load round trip, so that they do not occupy registers across the traversal (-2 % per
    // frame); the primary-only kernel is short enough that the round trip costs more than the two spills it saves.
    constexpr bool kLate = MODE != 0;
    float l_gl[3] = {0.0f, 0.0f, 0.0f}, l_scale = 0.0f;
    F3 l_eye{0.0f, 0.0f, 0.0f}, l_light{0.0f, 0.0f, 0.0f};
    LightSetup l_ls{};
    int l_hl[3] = {0, 0, 0}, l_dim = 0, l_shade_fast = 0;
    if constexpr (kLate) {
        const LateArgs la = late_args();
        const LateView lv_ = late_view();
#pragma unroll
        for (int k = 0; k < 3; ++k) { l_gl[k] = la->global_light[k]; l_hl[k] = la->highlighted[k]; }
        l_scale = la->voxel_scale;
        l_dim = la->tex_dim;
        l_shade_fast = la->shade_fast;
        l_light = F3{la->light_dir[0], la->light_dir[1], la->light_dir[2]};
        if constexpr (MODE == 1 && host_light<TRAV>::value) {
            l_ls.dir = l_light;
            l_ls.inv = F3{la->light_inv[0], la->light_inv[1], la->light_inv[2]};
            l_ls.push = F3{la->light_push[0], la->light_push[1], la->light_push[2]};
            l_ls.dposf = F3{la->light_dposf[0], la->light_dposf[1], la->light_dposf[2]};
            l_ls.dpos = I3{la->light_dpos[0], la->light_dpos[1], la->light_dpos[2]};
        }
        l_eye = F3{lv_->cam_pos[0], lv_->cam_pos[1], lv_->cam_pos[2]};
        lo = late_out(la, lv_);
    }
    // late copies where they were loaded, the arguments themselves (read where they are used) otherwise
    const auto gl_ = [&](int k) { if constexpr (kLate) return l_gl[k]; else return a.global_light[k]; };
    const auto hl_ = [&](int k) { if constexpr (kLate) return l_hl[k]; else return a.highlighted[k]; };
    const float gl[3] = {gl_(0), gl_(1), gl_(2)};
    const auto scale_ = [&]() { if constexpr (kLate) return l_scale; else return a.voxel_scale; };
    const auto dim_ = [&]() { if constexpr (kLate) return l_dim; else return a.tex_dim; };
    const bool shade_fast = (kLate ? l_shade_fast : a.shade_fast) != 0;
    const auto eye_ = [&]() { if constexpr (kLate) return l_eye; else return ray_origin; };  // ray_origin; gro = ray_origin * u_voxelScale
    const auto light_ = [&]() { if constexpr (kLate) return l_light; else return F3{a.light_dir[0], a.light_dir[1], a.light_dir[2]}; };
    float tc[3] = {gl[0], gl[1], gl[2]};
    if (!hit) {
        // distanceInMedium is still 0 here, so the absorption branch (comp:482) cannot fire
#pragma unroll
        for (int k = 0; k < 3; ++k) fc[k] = fc[k] + gl[k] * sky[k] * tc[k] * 1.0f;
    } else {
        // normal = length(hitNormal) > 0 ? hitNormal : (0,1,0)  (comp:497); hitNormal is n on h.axis
        int naxis = h.axis;
        float nval = h.n;
        if (!(__builtin_fabsf(h.n) > 0.0f)) { naxis = 1; nval = 1.0f; }
        // x / 1.0f == x: the divisions by u_voxelScale only cost instructions at the reference's scale of 1
        F3 hpw = h.point;
        if (scale_() != 1.0f) hpw = F3{h.point.x / scale_(), h.point.y / scale_(), h.point.z / scale_()};
        // distanceInMedium only feeds the absorption term, which needs mediumDensity > 0 (comp:501,512)
        float dist_in_medium = 0.0f;
        if (medium_density > 0.0f) dist_in_medium = 0.0f + len3(sub3(hpw, scale3(eye_(), scale_()))) / scale_();
        Decoded hv = decode_leaf(h.h0, h.h1);
        if (hv.c[3] <= 0.0f) { hv.p[0] = 1.0f; hv.p[1] = 0.0f; hv.p[2] = 0.0f; }
        float sc[4] = {hv.c[0], hv.c[1], hv.c[2], hv.c[3]};
        uint32_t sc_bytes = h.h0 & 0x00ffffffu;
        // surfaceColor = hitVoxel.color.a > 0 ? hitVoxel.color : lastVoxel.color (comp:505-507): the previous voxel is
        // decoded only in waves where some lane hit a voxel of alpha 0 (a phantom leaf); of its fields only the colour is
        // read on this path
        if (__builtin_amdgcn_ballot_w64((h.h0 >> 24) == 0u) != 0ull) {
            const Decoded lv = decode_leaf(h.p0, h.p1);
            if (!(hv.c[3] > 0.0f)) { sc[0] = lv.c[0]; sc[1] = lv.c[1]; sc[2] = lv.c[2]; sc[3] = lv.c[3]; sc_bytes = h.p0 & 0x00ffffffu; }
        }
        if (dist_in_medium > 1e-6f && medium_density > 0.0f) {  // comp:512-516
            float kk = -medium_density * dist_in_medium;
#pragma unroll
            for (int k = 0; k < 3; ++k) tc[k] = tc[k] * det_expf(kk * (1.0f - mc[k]));
        }
        bool inverted = false;
        if (h.map.x == hl_(0) && h.map.y == hl_(1) && h.map.z == hl_(2)) {
            sc[0] = 1.0f - sc[0]; sc[1] = 1.0f - sc[1]; sc[2] = 1.0f - sc[2]; sc[3] = 1.0f;
            inverted = true;
        }
        // dot products with an axis normal: (a*0 + b*n) + c*0 == b*n up to the sign of a zero, which neither
        // the comparisons nor the final rgba8 rounding can see (comp:522-524,537)
        float cosi = comp(ray_dir, naxis) * nval;
        const bool flipped = cosi > 0.0f;
        if (cosi > 0.0f) nval = -nval;
        F3 normal{naxis == 0 ? nval : 0.0f, naxis == 1 ? nval : 0.0f, naxis == 2 ? nval : 0.0f};
        const F3 light = light_();
        float ndotl = fmax_c(nval * comp(light, naxis), 0.0f);
        if (sc[3] >= 1.0f) {  // depth 0, first hit (comp:539-544)
            const int dim = dim_();
            int lin = h.map.x + dim * (h.map.y + dim * h.map.z);
            voxel_id = lin * 6 + face_index(h.axis, h.n);
            // (int)sqrt: the in-range form is the full one from 2^-96 up and returns 0 for 0; between them both round to 0
            const F3 dv = sub3(hpw, eye_());
            pixel_dist = (int)sqrt_inrange(dot3(dv, dv));
        }
        if (sc[3] < 1.0f) {  // translucent first hit: direct-lit fallback (comp:548-553)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float direct = gl[k] * ndotl;
                float lit = sc[k] * direct;
                fc[k] = fc[k] + tc[k] * lit * 1.0f;
            }
        } else {
            float emission = hv.p[1] * 10.0f;
            if (emission > 0.0f) {  // comp:575-578
#pragma unroll
                for (int k = 0; k < 3; ++k) fc[k] = fc[k] + tc[k] * sc[k] * emission * 1.0f;
            } else {
                int lit = 1;
                if constexpr (MODE == 1 && host_light<TRAV>::value) lit = TRAV::shadow(a, tc_, add3(h.point, scale3(normal, 2e-3f)), l_ls, h);
                else if constexpr (MODE == 1) lit = TRAV::shadow(a, tc_, add3(h.point, scale3(normal, 2e-3f)), light, h);
                if constexpr (MODE == 1) {   // the diffuse bounce of comp:596-616 starts from here: what pass 2 needs of this hit
                    seed_word = kSeedValid | sc_bytes | ((uint32_t)naxis << 24) | (nval < 0.0f ? 1u << 26 : 0u) | (inverted ? 1u << 27 : 0u) |
                                (lit ? 1u << 28 : 0u);
                    // the bounce ray travels in the medium on this side of the surface: n1 of comp:519-524 (the previous voxel's index
                    // where it has one, else 1.0; the hit voxel's when the normal was flipped)
                    const float prev_r = (h.p0 >> 24) != 0u ? refraction_of(h.p1) : 0.0f;
                    const float n1 = prev_r > 0.0f ? prev_r : 1.0f, n2 = hv.p[0] > 0.0f ? hv.p[0] : 1.0f;
                    seed_iof = flipped ? n2 : n1;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float direct = gl[k] * (float)lit * ndotl;
                    const float num = direct * sc[k] * tc[k] * 1.0f;
                    fc[k] = fc[k] + (shade_fast ? div_pi_inrange(num) : num / kPI);
                }
            }
        }
    }
    rgba = unorm8(fc[0]) | (unorm8(fc[1]) << 8) | (unorm8(fc[2]) << 16) | (255u << 24);
    idd = make_int2(voxel_id, pixel_dist);
    if constexpr (MODE == 1) {
        if (seed_regs) { seed_regs->hp = h.point; seed_regs->word = seed_word; seed_regs->iof = seed_iof; }
        if (seed) {   // wave-uniform; every in-range pixel writes its word so that pass 2 never reads a stale one
            seed[3 * 64] = seed_word;
            if (seed_word) {
                seed[0] = __float_as_uint(h.point.x); seed[64] = __float_as_uint(h.point.y); seed[2 * 64] = __float_as_uint(h.point.z);
                seed[4 * 64] = __float_as_uint(seed_iof);
            }
        }
    }
}

namespace full {  // MODE 2 and 3 (3: the last diffuse bounce of a pixel goes to a queue for bounce_kernel), defined in vrt_full.hip.h
template <class TRAV, bool DEFER>
__device__ void trace_pixel_full(const KArgs &a, const View &vw, const typename TRAV::Ctx &tc_, int px, int py, uint32_t &rgba, int2 &idd, LateOut &lo,
                                 uint32_t queue, uint32_t out_offset);
// pass 2 of the two-pass form: the diffuse bounce of a seeded pixel; false when the pixel has none (rgba untouched)
template <class TRAV>
__device__ bool bounce_pixel(const KArgs &a, const typename TRAV::Ctx &tc_, int px, int py, Seed seed, uint32_t &rgba);
}

// One lane per pixel; a wave covers a TW x TH pixel tile (TW*TH == 64) so the 64 rays of a wave stay spatially
// coherent. One tile per wave (PERSIST: a fixed grid walks the tiles with a grid-stride loop instead).
// WPE: waves per SIMD the register allocator must leave room for (1 = no constraint beyond BLOCK). It also bounds
// the SCALAR registers: a SIMD admits floor(800 / (ceil(sgprs / 16) * 16 + 16)) waves (MI355X_MICROARCH.md,
// "Residency") -- 6 with the 106 this kernel takes when unconstrained, whatever its 67 vector registers would
// allow -- and amdgpu_waves_per_eu(7) makes the compiler stay within the 96 that admit 7.
// SCHED: feedback scheduling flavours (KArgs::group_order / tile_cost).
template <int MODE, class TRAV, int TW, int BLOCK, int WPE = 1, bool PERSIST = false, int SCHED = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WPE))) void trace_kernel(const KArgs a, const ViewSet vs) {
    extern __shared__ __attribute__((aligned(16))) uint2 lds_dyn[];
    constexpr int TH = 64 / TW;
    constexpr int WAVES = BLOCK / 64;
    typename TRAV::Ctx tc_;
    TRAV::template block_init<BLOCK>(a, lds_dyn, tc_);
    // only the traversals that stage records in LDS have anything to wait for: the others start tracing as soon
    // as the wave is launched, without meeting the other waves of the workgroup
    if constexpr (TRAV::kStagesLds) __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int tiles_x = (a.width + TW - 1) / TW;
    const int tiles_y = (a.n_rows + TH - 1) / TH;
    const int n_tiles = tiles_x * tiles_y;
    int lx = lane % TW, ly = lane / TW;
    // One tile per wave and no loop unless PERSIST: without the back edge the kernel arguments need not stay live
    // after ray generation, which is worth ~19 VGPRs (88 -> 69) and two thirds of the SGPR spills on gfx950.
    static_assert(!(PERSIST && SCHED), "the scheduled flavours trace one tile per wave");
    int first = blockIdx.x * WAVES;  // first tile of this workgroup
    bool part_wave = false;          // KArgs::split_count: this wave traces one row of its tile
    if constexpr (SCHED & 1) {
        static_assert(kGroupTiles % WAVES == 0, "a scheduling group is a whole number of workgroups");
        constexpr int kWgPerGroup = kGroupTiles / WAVES;
        // The general full path tracer, ordered launches that do not measure (KArgs::split_count): a frame of a translucent scene is as
        // long as its longest wave -- 79 rays one after the other behind the glass ball of the reference's room, every round as long
        // as the longest of the wave's 64 marches (profiles/r03_room_critical_path.txt) -- so the few heaviest groups are traced as
        // kSplitParts waves per tile, one row of 8 pixels each: the longest of 8 marches instead of 64 per round, 23 % off that wave.
        constexpr bool kSplit = MODE == 2 && WAVES == 1 && TW == 8;   // SCHED 1 and 3: a measuring launch under an order splits like the others
        uint32_t wg = blockIdx.x;
        if constexpr (kSplit) {
            const uint32_t n_split = a.split_count ? *a.split_count : 0u;
            constexpr uint32_t kPerGroup = (uint32_t)(kGroupTiles * kSplitParts);
            if (wg < n_split * kPerGroup) {
                part_wave = true;
                first = (int)a.group_order[wg / kPerGroup] * kGroupTiles + (int)((wg % kPerGroup) / (uint32_t)kSplitParts);
                if (lane >= 64 / kSplitParts) return;
                const int at = (int)(wg % (uint32_t)kSplitParts) * (64 / kSplitParts) + lane;   // the part's pixels in the tile's row-major order
                lx = at % TW;
                ly = at / TW;
            } else {
                wg = wg - n_split * kPerGroup + n_split * (uint32_t)kGroupTiles;   // its place among the whole-tile workgroups
                if (wg >= (uint32_t)((n_tiles + kGroupTiles - 1) / kGroupTiles * kGroupTiles)) return;   // a workgroup no split group needed
            }
        }
        if (!part_wave) first = (int)a.group_order[wg / kWgPerGroup] * kGroupTiles + (int)(wg % kWgPerGroup) * WAVES;
    }
    unsigned long long t_begin = 0;
    if constexpr (SCHED & 2) t_begin = __builtin_readcyclecounter();
    for (int tile = first + wave; tile < n_tiles; tile += gridDim.x * WAVES) {
        int tx, ty;
        if (a.tiles_x_magic) {
            ty = (int)__umulhi((uint32_t)tile, a.tiles_x_magic);
            tx = tile - ty * tiles_x;
        } else {
            tx = tile % tiles_x;
            ty = tile / tiles_x;
        }
        int px = tx * TW + lx;
        int j = ty * TH + ly;
        if (px < a.width && j < a.n_rows) {
            int py;
            if (a.row_mode == 1) py = a.row0 + j;
            else if (a.row_mode == 2 && TH == 8) py = a.row0 + ty * a.row_stride + ly;
            else py = a.row0 + (j / a.tile_rows) * a.row_stride + (j % a.tile_rows);
            uint32_t rgba;
            int2 idd;
            const View &vw = vs.v[blockIdx.y];
            LateOut lo;  // MODE 1, 2, 3: the output side of the arguments, re-read after the trace rather than kept in registers across it
            lo.skip_rgba = false;
            if constexpr (MODE == 5) {   // pass 2 of the two-pass full path tracer: only the colour of seeded pixels is (re)written
                const uint32_t *sp = reinterpret_cast<const uint32_t *>(a.defer_rec) + ((size_t)tile * kSeedPlanes) * 64 + lane;
                Seed seed;
                seed.word = sp[3 * 64];
                seed.hp = F3{0.0f, 0.0f, 0.0f};
                seed.iof = 1.0f;
                if (seed.word & kSeedValid) {
                    seed.hp = F3{__uint_as_float(sp[0]), __uint_as_float(sp[64]), __uint_as_float(sp[2 * 64])};
                    seed.iof = __uint_as_float(sp[4 * 64]);
                }
                if (full::bounce_pixel<TRAV>(a, tc_, px, py, seed, rgba)) {
                    const LateArgs la = late_args();
                    const LateView lv_ = late_view();
                    lv_->out_rgba[(size_t)(la->compact ? j : py) * (size_t)la->width + (size_t)px] = rgba;
                }
            } else {
            if constexpr (MODE == 2 || MODE == 3)
                full::trace_pixel_full<TRAV, MODE == 3>(a, vw, tc_, px, py, rgba, idd, lo, (uint32_t)tile % kDeferQueues,
                                                        (uint32_t)((a.compact ? j : py) * a.width + px));
            else if constexpr (MODE == 4)   // pass 1: the primary + shadow kernel, leaving a seed per pixel
                trace_pixel<1, TRAV>(a, vw, tc_, px, py, rgba, idd, lo, reinterpret_cast<uint32_t *>(a.defer_rec) + ((size_t)tile * kSeedPlanes) * 64 + lane);
            else if constexpr (MODE == 6) {   // both passes in this wave: the seed stays in registers
                Seed seed;
                seed.word = 0u;
                trace_pixel<1, TRAV>(a, vw, tc_, px, py, rgba, idd, lo, nullptr, &seed);
                uint32_t both;
                if (full::bounce_pixel<TRAV>(a, tc_, px, py, seed, both)) rgba = both;
            }
            else trace_pixel<MODE, TRAV>(a, vw, tc_, px, py, rgba, idd, lo);
            if constexpr (MODE == 0) lo = LateOut{vw.out_rgba, vw.out_id, a.width, a.compact, false};
            size_t o = (size_t)(lo.compact ? j : py) * (size_t)lo.width + (size_t)px;
            if (lo.out_rgba && !lo.skip_rgba) lo.out_rgba[o] = rgba;
            if (lo.out_id) lo.out_id[o] = idd;
            }
        }
        if constexpr (SCHED & 2) {  // the wave has reconverged: this is the time its slowest ray took
#ifndef VRT_EXP_STATS
            // a tile traced as part-tile waves reports the longest of them scaled to what the whole tile would have taken (x 21/16: the
            // room's heaviest row alone is 1.00 ms as whole tiles, 0.77 as eight parts), so that it keeps its place in the next order;
            // the dispatcher zeroes the ticks before such a launch
            if (part_wave) { if (lane == 0) atomicMax(&a.tile_cost[tile], (uint32_t)(((__builtin_readcyclecounter() - t_begin) * 21ull) >> 4)); }
            else if (lane == 0) a.tile_cost[tile] = (uint32_t)(__builtin_readcyclecounter() - t_begin);
#endif
        }
        if constexpr (!PERSIST) break;
    }
}

}  // namespace vrt
