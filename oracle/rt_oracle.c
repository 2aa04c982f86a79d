/*
 * rt_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Scalar CPU restatement of shaders/raytracing.comp, one pixel at a time, in
 * strict IEEE-754 binary32 (round-to-nearest-even, no FMA contraction -- build
 * with -ffp-contract=off, no -ffast-math).
 *
 * PARITY UNPINNED for the shader arithmetic: the reference ships no test,
 * golden image or known-answer vector for raytracing.comp and GLSL cannot be
 * executed in the build container. Where GLSL leaves precision or evaluation
 * order implementation-defined this file fixes a convention (listed here) and
 * the HIP kernels are held bit-exact to THAT:
 *   C1 mat4*vec4        = (m0*x + m1*y) + (m2*z + m3*w)           (glm order)
 *   C2 dot(a,b) vec3    = (ax*bx + ay*by) + az*bz
 *   C3 inversesqrt(x)   = 1.0f / sqrtf(x); normalize(v) = v * inversesqrt(dot(v,v))
 *   C4 length(v)        = sqrtf(dot(v,v))
 *   C5 a / b, sqrt      = correctly rounded
 *   C6 min(a,b)         = b < a ? b : a ; max(a,b) = a < b ? b : a   (GLSL spec text)
 *   C7 rgba8 imageStore = rint(clamp(c,0,1) * 255) ties-to-even
 *   C8 out-of-world octreeFind early return (comp:143-145) leaves nodeMin/Max/
 *      nodeCoord undefined in GLSL; here: world bounds / 0
 *   C9 exp/sin/cos/pow  = the o_det_* polynomial routines below (full mode and
 *      the in-medium absorption term only)
 *   C10 float(uint) in rand() rounds to nearest-even (may yield 1.0)
 * Modes (subset selection of pathTrace, comp:435-622):
 *   PRIMARY        : primary ray only; the shadow factor of comp:587 is taken as 1;
 *                    no secondary rays are pushed; a translucent first hit
 *                    (surfaceColor.a < 1) takes the direct-lit fallback of comp:548-553.
 *   PRIMARY_SHADOW : PRIMARY + notInShadow (comp:333-377) for the direct term.
 *   FULL           : the whole shader (glass stack, diffuse bounce, RNG).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { int32_t x, y, z; } i3;

typedef struct {          /* comp:45-51 VoxelData */
    float color[4];
    float props[3];
    i3 nmin, nmax;
    int32_t coord;        /* linear texel index (== toLinear(nodeCoord)) */
} vox_t;

typedef struct {          /* comp:57-68 Ray */
    v3 origin, dir;
    float iof, weight;
    int defined;
    float tint[4];
    float dist_in_medium;
    float medium_color[4];
    float medium_density;
    int depth;
} ray_t;

typedef struct {
    const o_scene *s;
    o_stats st;
    uint32_t px_fetches;
    uint32_t rng;
    uint32_t px_index;   /* y*W + x of the pixel being traced (find trace only) */
    int par_depth;       /* depth of the cached parent (find trace only) */
} ctx_t;

/* optional find trace for traversal studies (tools/): one record per octreeFind call inside the world */
static o_find_rec *g_trace = NULL;
static size_t g_trace_cap = 0, g_trace_n = 0;
void o_set_find_trace(o_find_rec *buf, size_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
size_t o_find_trace_count(void) { return g_trace_n; }

#define MAX_RAYS 8
#define BOUNCES 1
#define INDIRECT_SAMPLES 1
static const float kPI = 3.14159265359f;
static const float kSky[3] = {0.5f, 0.7f, 1.0f};
static const float kSun = 3.0f;

/* ---- conventions ---------------------------------------------------------- */
static inline float fmin_c(float a, float b) { return b < a ? b : a; }
static inline float fmax_c(float a, float b) { return a < b ? b : a; }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 scale3(v3 a, float s) { v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline v3 add3(v3 a, v3 b) { v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline v3 normalize3(v3 a) { return scale3(a, 1.0f / sqrtf(dot3(a, a))); }
static inline float sign_c(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline v3 cross3(v3 x, v3 y) {
    v3 r = {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y};
    return r;
}

/* ---- deterministic transcendental conventions (C9) ------------------------- */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

float o_det_expf(float x) {
    if (x > 88.0f) return bits2f(0x7f800000u);
    if (x < -87.0f) return 0.0f;
    float k = rintf(x * 1.44269504088896341f);
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
    return p * bits2f((uint32_t)(ki + 127) << 23);
}

static float det_logf(float x) { /* x > 0, normal */
    uint32_t u = f2bits(x);
    int e = (int)(u >> 23) - 126;
    float m = bits2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
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

float o_det_powf(float x, float y) {
    if (x <= 0.0f) return 0.0f;
    if (x < 1.17549435e-38f) return 0.0f;
    return o_det_expf(y * det_logf(x));
}

/* shared octant reduction for sin/cos, x >= 0 expected (|x| used) */
static void det_sincos(float xin, float *s_out, float *c_out) {
    float x = fabsf(xin);
    int sign_s = xin < 0.0f ? -1 : 1, sign_c = 1;
    int j = (int)(x * 1.27323954473516f); /* 4/pi */
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
    *s_out = sign_s < 0 ? -sv : sv;
    *c_out = sign_c < 0 ? -cv : cv;
}
float o_det_sinf(float x) { float s, c; det_sincos(x, &s, &c); return s; }
float o_det_cosf(float x) { float s, c; det_sincos(x, &s, &c); return c; }

/* ---- texel access --------------------------------------------------------- */
/* comp:132-135 getNodeData: texelFetch on the zero-padded volume */
static inline uint32_t fetch(ctx_t *c, int32_t idx) {
    c->st.fetches++;
    c->px_fetches++;
    if (idx < 0 || (size_t)idx >= c->s->n_texels) return 0u;
    const uint8_t *p = c->s->texels + (size_t)idx * 4;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

static inline int in_world(const o_scene *s, i3 p) { /* comp:224-226 */
    return p.x >= s->bounds_min[0] && p.y >= s->bounds_min[1] && p.z >= s->bounds_min[2] &&
           p.x < s->bounds_max[0] && p.y < s->bounds_max[1] && p.z < s->bounds_max[2];
}

static inline int popc8(uint32_t m) { int n = 0; while (m) { m &= m - 1; n++; } return n; }

/* comp:137-220 octreeFind */
static vox_t octree_find(ctx_t *c, i3 wp, i3 *min_b, i3 *max_b, int32_t *cur_coord) {
    const o_scene *s = c->s;
    vox_t d;
    memset(&d, 0, sizeof d);
    c->st.finds++;
    if (!in_world(s, wp)) { /* C8 */
        d.nmin.x = s->bounds_min[0]; d.nmin.y = s->bounds_min[1]; d.nmin.z = s->bounds_min[2];
        d.nmax.x = s->bounds_max[0]; d.nmax.y = s->bounds_max[1]; d.nmax.z = s->bounds_max[2];
        return d;
    }
    int inside = wp.x >= min_b->x && wp.y >= min_b->y && wp.z >= min_b->z &&
                 wp.x < max_b->x && wp.y < max_b->y && wp.z < max_b->z;
    int depth = (inside && *cur_coord != 0) ? c->par_depth : 0, start_depth = depth; /* coord 0 = root */
    if (inside) { d.coord = *cur_coord; d.nmin = *min_b; d.nmax = *max_b; }
    else {
        c->st.root_restarts++;
        d.coord = 0;
        d.nmin.x = s->bounds_min[0]; d.nmin.y = s->bounds_min[1]; d.nmin.z = s->bounds_min[2];
        d.nmax.x = s->bounds_max[0]; d.nmax.y = s->bounds_max[1]; d.nmax.z = s->bounds_max[2];
    }
    int is_leaf = 0;
    for (int i = 0; i < 16; i++) {
        uint32_t nd = fetch(c, d.coord);
        if (is_leaf) {
            if (g_trace && g_trace_n < g_trace_cap) {
                o_find_rec *r = &g_trace[g_trace_n++];
                r->pixel = c->px_index; r->x = (int16_t)wp.x; r->y = (int16_t)wp.y; r->z = (int16_t)wp.z;
                r->found_depth = (uint8_t)depth; r->start_depth = (uint8_t)start_depth; r->leaf = 1;
            }
            uint32_t pd = fetch(c, d.coord + 1);
            d.color[0] = (float)(nd & 0xff) / 255.0f;
            d.color[1] = (float)((nd >> 8) & 0xff) / 255.0f;
            d.color[2] = (float)((nd >> 16) & 0xff) / 255.0f;
            d.color[3] = (float)((pd >> 24) & 0xff) / 255.0f;
            float pr = (float)(pd & 0xff) / 255.0f;
            float pg = (float)((pd >> 8) & 0xff) / 255.0f;
            float pb = (float)((pd >> 16) & 0xff) / 255.0f;
            d.props[0] = pr * 3.0f; d.props[1] = pg; d.props[2] = pb; /* comp:126-128 */
            return d;
        }
        uint32_t base = s->wide_pointers ? (uint32_t)d.coord + 1u : (nd & 0x7fffffu);   /* comp:89-96; EXT: see oracle.h */
        i3 mid = {d.nmin.x + (d.nmax.x - d.nmin.x) / 2, d.nmin.y + (d.nmax.y - d.nmin.y) / 2,
                  d.nmin.z + (d.nmax.z - d.nmin.z) / 2};
        int ci = (wp.x >= mid.x ? 4 : 0) + (wp.y >= mid.y ? 2 : 0) + (wp.z >= mid.z ? 1 : 0);
        uint32_t mask = nd >> 24;
        int exists = (mask >> ci) & 1u;
        uint32_t off = (uint32_t)popc8(mask & ((1u << ci) - 1u));
        uint32_t ptr = fetch(c, (int32_t)(base + off));  /* fetched even when absent, comp:196-199 */
        uint32_t next = (ptr & 0x7fffffu) | (s->wide_pointers ? (ptr >> 24) << 23 : 0u);
        is_leaf = (ptr & 0x800000u) != 0;
        *cur_coord = d.coord; *min_b = d.nmin; *max_b = d.nmax;
        c->par_depth = depth;
        depth++;
        d.coord = (int32_t)next;
        if (ci & 4) d.nmin.x = mid.x; else d.nmax.x = mid.x;  /* comp:105-118 */
        if (ci & 2) d.nmin.y = mid.y; else d.nmax.y = mid.y;
        if (ci & 1) d.nmin.z = mid.z; else d.nmax.z = mid.z;
        if (!exists) {
            if (g_trace && g_trace_n < g_trace_cap) {
                o_find_rec *r = &g_trace[g_trace_n++];
                r->pixel = c->px_index; r->x = (int16_t)wp.x; r->y = (int16_t)wp.y; r->z = (int16_t)wp.z;
                r->found_depth = (uint8_t)depth; r->start_depth = (uint8_t)start_depth; r->leaf = 0;
            }
            d.color[0] = d.color[1] = d.color[2] = d.color[3] = 0.0f;
            return d;
        }
    }
    return d;
}

static inline i3 floor_i3(v3 p) { i3 r = {(int32_t)floorf(p.x), (int32_t)floorf(p.y), (int32_t)floorf(p.z)}; return r; }
static inline float vget(v3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

/* comp:248-330 hitMarching */
static int hit_marching(ctx_t *c, v3 origin, v3 dir, float ray_iof, i3 *hit_map, v3 *hit_point,
                        v3 *hit_normal, vox_t *prev, vox_t *hit) {
    const o_scene *s = c->s;
    v3 rp = origin;
    float inv_len = 1.0f / sqrtf(dot3(dir, dir));
    dir = scale3(dir, inv_len);
    v3 inv;
    inv.x = (fabsf(dir.x) < 1e-8f) ? 1e20f : 1.0f / dir.x;
    inv.y = (fabsf(dir.y) < 1e-8f) ? 1e20f : 1.0f / dir.y;
    inv.z = (fabsf(dir.z) < 1e-8f) ? 1e20f : 1.0f / dir.z;
    int32_t cur = 0;
    i3 nmin = {s->bounds_min[0], s->bounds_min[1], s->bounds_min[2]};
    i3 nmax = {s->bounds_max[0], s->bounds_max[1], s->bounds_max[2]};
    i3 mp = floor_i3(rp);
    *hit = octree_find(c, mp, &nmin, &nmax, &cur);
    *prev = *hit;
    hit_normal->x = hit_normal->y = hit_normal->z = 0.0f;
    for (int i = 0; i < 1024; i++) {
        c->st.steps++;
        v3 bmin = {(float)hit->nmin.x, (float)hit->nmin.y, (float)hit->nmin.z};
        v3 bmax = {(float)hit->nmax.x, (float)hit->nmax.y, (float)hit->nmax.z};
        v3 tp;
        tp.x = (dir.x > 0.0f ? bmax.x : bmin.x) - rp.x;
        tp.y = (dir.y > 0.0f ? bmax.y : bmin.y) - rp.y;
        tp.z = (dir.z > 0.0f ? bmax.z : bmin.z) - rp.z;
        v3 tm = {tp.x * inv.x, tp.y * inv.y, tp.z * inv.z};
        float t = fmin_c(tm.x, fmin_c(tm.y, tm.z));
        int axis = (tm.x < tm.y) ? ((tm.x < tm.z) ? 0 : 2) : ((tm.y < tm.z) ? 1 : 2);
        float n = -sign_c(vget(dir, axis));
        hit_normal->x = axis == 0 ? n : 0.0f;
        hit_normal->y = axis == 1 ? n : 0.0f;
        hit_normal->z = axis == 2 ? n : 0.0f;
        rp.x = rp.x + dir.x * t; rp.y = rp.y + dir.y * t; rp.z = rp.z + dir.z * t;
        float push = sign_c(vget(dir, axis)) * 0.0001f;
        if (axis == 0) rp.x = rp.x + push; else if (axis == 1) rp.y = rp.y + push; else rp.z = rp.z + push;
        mp = floor_i3(rp);
        if (!in_world(s, mp)) return 0;
        *prev = *hit;
        *hit = octree_find(c, mp, &nmin, &nmax, &cur);
        float pr = (prev->color[3] > 0.0f && prev->props[0] > 0.0f) ? prev->props[0] : ray_iof;
        float cr = (hit->color[3] > 0.0f && hit->props[0] > 0.0f) ? hit->props[0] : 1.0f;
        if (fabsf(cr - pr) > 0.0001f) { *hit_map = mp; *hit_point = rp; return 1; }
    }
    return 0;
}

/* comp:333-377 notInShadow (lightDir is used as given, not re-normalised) */
static int not_in_shadow(ctx_t *c, v3 origin, v3 ld) {
    const o_scene *s = c->s;
    c->st.shadow_rays++;
    v3 rp = origin, inv;
    inv.x = (fabsf(ld.x) < 1e-8f) ? 1e20f : 1.0f / ld.x;
    inv.y = (fabsf(ld.y) < 1e-8f) ? 1e20f : 1.0f / ld.y;
    inv.z = (fabsf(ld.z) < 1e-8f) ? 1e20f : 1.0f / ld.z;
    i3 mp = floor_i3(rp);
    int32_t cur = 0;
    i3 nmin = {s->bounds_min[0], s->bounds_min[1], s->bounds_min[2]};
    i3 nmax = {s->bounds_max[0], s->bounds_max[1], s->bounds_max[2]};
    for (int i = 0; i < 64; i++) {
        c->st.steps++;
        vox_t v = octree_find(c, mp, &nmin, &nmax, &cur);
        if (v.color[3] > 0.1f && v.props[1] == 0.0f) return 0;
        v3 tp;
        tp.x = (ld.x > 0.0f ? (float)v.nmax.x : (float)v.nmin.x) - rp.x;
        tp.y = (ld.y > 0.0f ? (float)v.nmax.y : (float)v.nmin.y) - rp.y;
        tp.z = (ld.z > 0.0f ? (float)v.nmax.z : (float)v.nmin.z) - rp.z;
        v3 tm = {tp.x * inv.x, tp.y * inv.y, tp.z * inv.z};
        float t = fmin_c(tm.x, fmin_c(tm.y, tm.z));
        int axis = (tm.x < tm.y) ? ((tm.x < tm.z) ? 0 : 2) : ((tm.y < tm.z) ? 1 : 2);
        rp.x = rp.x + ld.x * t; rp.y = rp.y + ld.y * t; rp.z = rp.z + ld.z * t;
        float push = sign_c(vget(ld, axis)) * 0.001f;
        if (axis == 0) rp.x = rp.x + push; else if (axis == 1) rp.y = rp.y + push; else rp.z = rp.z + push;
        mp = floor_i3(rp);
        if (!in_world(s, mp)) return 1;
    }
    return 1;
}

/* comp:381-399 */
static void init_rng(ctx_t *c, int px, int py, int sample) {
    uint32_t seed = (uint32_t)px + (uint32_t)py * 1920u + 123456u + (uint32_t)sample * 78901u;
    uint32_t st = seed * 747796405u + 2891336453u;
    uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    c->rng = (w >> 22u) ^ w;
}
static float rand_f(ctx_t *c) {
    c->rng = c->rng * 747796405u + 2891336453u;
    uint32_t w = ((c->rng >> ((c->rng >> 28u) + 4u)) ^ c->rng) * 277803737u;
    c->rng = (w >> 22u) ^ w;
    return (float)c->rng / 4294967296.0f; /* C10 */
}

/* comp:402-417 */
static v3 cosine_hemisphere(v3 n, float rx, float ry) {
    float phi = 2.0f * kPI * ry;
    float ct = sqrtf(rx);
    float stheta = sqrtf(1.0f - rx);
    float x = stheta * o_det_cosf(phi);
    float z = stheta * o_det_sinf(phi);
    v3 up;
    if (fabsf(n.z) < 0.999f) { up.x = 0.0f; up.y = 0.0f; up.z = 1.0f; } else { up.x = 1.0f; up.y = 0.0f; up.z = 0.0f; }
    v3 tangent = normalize3(cross3(up, n));
    v3 bitangent = cross3(n, tangent);
    v3 r = add3(add3(scale3(tangent, x), scale3(bitangent, z)), scale3(n, ct));
    return normalize3(r);
}

/* comp:419-433 */
static int face_index(v3 n) {
    if (len3(n) < 0.5f) return 0;
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax > ay && ax > az) return n.x > 0.0f ? 0 : 1;
    else if (ay > az) return n.y > 0.0f ? 2 : 3;
    else return n.z > 0.0f ? 4 : 5;
}

/* GLSL refract / reflect (spec 8.5) */
static v3 refract3(v3 I, v3 N, float eta) {
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    v3 z = {0.0f, 0.0f, 0.0f};
    if (k < 0.0f) return z;
    float f = eta * d + sqrtf(k);
    return sub3(scale3(I, eta), scale3(N, f));
}
static v3 reflect3(v3 I, v3 N) { float d = dot3(N, I); return sub3(I, scale3(N, 2.0f * d)); }

static void absorb(float *tc, float density, float dist, const float *mc) { /* comp:482-486,512-516 */
    float k = -density * dist;
    tc[0] = tc[0] * o_det_expf(k * (1.0f - mc[0]));
    tc[1] = tc[1] * o_det_expf(k * (1.0f - mc[1]));
    tc[2] = tc[2] * o_det_expf(k * (1.0f - mc[2]));
}

static ray_t make_ray(v3 o, v3 d, float iof, float w, const float *tint, float dim, const float *mc, float md, int depth) {
    ray_t r;
    r.origin = o; r.dir = d; r.iof = iof; r.weight = w; r.defined = 1;
    memcpy(r.tint, tint, 16); r.dist_in_medium = dim; memcpy(r.medium_color, mc, 16);
    r.medium_density = md; r.depth = depth;
    return r;
}

/* comp:435-622 pathTrace */
static void path_trace(ctx_t *c, v3 ray_origin, v3 ray_dir, int mode, float out_rgb[3], int32_t *voxel_id, int32_t *pixel_dist) {
    const o_scene *s = c->s;
    int32_t cur = 0;
    i3 nmin = {s->bounds_min[0], s->bounds_min[1], s->bounds_min[2]};
    i3 nmax = {s->bounds_max[0], s->bounds_max[1], s->bounds_max[2]};
    *voxel_id = 0;
    *pixel_dist = s->bounds_max[0] - s->bounds_min[0]; /* worldSize.x, comp:34,441 */
    v3 gro = scale3(ray_origin, s->voxel_scale);
    i3 this_mp = floor_i3(gro);
    vox_t tv = octree_find(c, this_mp, &nmin, &nmax, &cur);
    float start_iof = (tv.props[0] > 0.0f && tv.props[0] < 3.0f) ? tv.props[0] : 1.0f;

    ray_t stack[MAX_RAYS];
    for (int i = 0; i < MAX_RAYS; i++) stack[i].defined = 0;
    float inv_len = 1.0f / sqrtf(dot3(ray_dir, ray_dir));
    ray_dir = scale3(ray_dir, inv_len);
    const float ones[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    stack[0] = make_ray(gro, ray_dir, start_iof, 1.0f, s->global_light, 0.0f,
                        tv.color[3] > 0.0f ? tv.color : ones, tv.color[3] * 5.0f, 0);
    int sp = 1;
    float fc[3] = {0.0f, 0.0f, 0.0f};
    const float *gl = s->global_light;
    v3 light = {s->light_dir[0], s->light_dir[1], s->light_dir[2]};

    while (sp > 0) {
        ray_t r = stack[--sp];
        stack[sp].defined = 0;
        if (!r.defined) continue;
        i3 mp = {0, 0, 0};
        v3 hp = {0, 0, 0}, hn = {0, 0, 0};
        vox_t last, hv;
        int hit = hit_marching(c, r.origin, r.dir, r.iof, &mp, &hp, &hn, &last, &hv);
        float tc[4] = {r.tint[0], r.tint[1], r.tint[2], r.tint[3]};
        if (!hit && r.depth <= 0) {
            if (r.dist_in_medium > 1e-6f && r.medium_density > 0.0f) absorb(tc, r.medium_density, r.dist_in_medium, r.medium_color);
            for (int k = 0; k < 3; k++) fc[k] = fc[k] + gl[k] * kSky[k] * tc[k] * r.weight;
            continue;
        } else if (!hit) {
            for (int k = 0; k < 3; k++) fc[k] = fc[k] + tc[k] * kSky[k] * kSun * r.weight / kPI;
            continue;
        }
        if (r.depth == 0) c->st.hits++;
        v3 normal = hn;
        if (!(len3(hn) > 0.0f)) { normal.x = 0.0f; normal.y = 1.0f; normal.z = 0.0f; }
        v3 hpw = {hp.x / s->voxel_scale, hp.y / s->voxel_scale, hp.z / s->voxel_scale};
        r.dist_in_medium = r.dist_in_medium + len3(sub3(hpw, r.origin)) / s->voxel_scale;
        if (hv.color[3] <= 0.0f) { hv.props[0] = 1.0f; hv.props[1] = 0.0f; hv.props[2] = 0.0f; }
        if (last.color[3] <= 0.0f) {
            if (r.iof > 0.0f) { last.props[0] = last.props[1] = last.props[2] = 0.0f; }
            else { last.props[0] = 1.0f; last.props[1] = 0.0f; last.props[2] = 0.0f; }
        }
        float sc[4];
        memcpy(sc, hv.color[3] > 0.0f ? hv.color : last.color, 16);
        float n2 = hv.props[0] > 0.0f ? hv.props[0] : 1.0f;
        float n1 = last.props[0] > 0.0f ? last.props[0] : 1.0f;
        v3 inc = r.dir;
        if (r.dist_in_medium > 1e-6f && r.medium_density > 0.0f) absorb(tc, r.medium_density, r.dist_in_medium, r.medium_color);
        if (mp.x == s->highlighted[0] && mp.y == s->highlighted[1] && mp.z == s->highlighted[2]) {
            sc[0] = 1.0f - sc[0]; sc[1] = 1.0f - sc[1]; sc[2] = 1.0f - sc[2]; sc[3] = 1.0f;
        }
        float cosi = dot3(inc, normal);
        if (cosi > 0.0f) { normal.x = -normal.x; normal.y = -normal.y; normal.z = -normal.z; float t = n1; n1 = n2; n2 = t; }
        float ndotl = fmax_c(dot3(normal, light), 0.0f);

        if (r.depth == 0 && *voxel_id == 0 && sc[3] >= 1.0f) { /* comp:539-544 */
            int32_t lin = mp.x + s->tex_dim * (mp.y + s->tex_dim * mp.z);
            *voxel_id = lin * 6 + face_index(hn);
            *pixel_dist = (int32_t)len3(sub3(hpw, ray_origin));
        }

        if (r.depth <= 0 && sc[3] < 1.0f) { /* translucent, comp:547-572 */
            float reflect_i = 0.0f, refract_i = 0.0f;
            int has_tir = 0;
            v3 refr_dir = {0, 0, 0};
            if (mode == O_MODE_FULL) {
                refr_dir = refract3(inc, normal, n1 / n2);
                float R0 = (n1 - n2) / (n1 + n2) * (n1 - n2) / (n1 + n2);
                v3 ninc = {-inc.x, -inc.y, -inc.z};
                float cos_t = fmax_c(0.0f, dot3(ninc, normal));
                float fres = R0 + (1.0f - R0) * o_det_powf(1.0f - cos_t, 5.0f);
                fres = fmin_c(fmax_c(fres, 0.0f), 1.0f); /* clamp = min(max(x,lo),hi) */
                has_tir = len3(refr_dir) < 0.001f;
                reflect_i = fres;
                refract_i = has_tir ? 0.0f : (1.0f - fres);
            }
            if (mode != O_MODE_FULL || sp == MAX_RAYS || reflect_i <= 0.001f || refract_i <= 0.001f) {
                for (int k = 0; k < 3; k++) {
                    float direct = gl[k] * ndotl;
                    float lit = sc[k] * direct;
                    fc[k] = fc[k] + tc[k] * lit * r.weight;
                }
                continue;
            }
            if (reflect_i > 0.001f && sp < MAX_RAYS) {
                float rw = r.weight * reflect_i;
                if (rw > 1e-4f)
                    stack[sp++] = make_ray(add3(hp, scale3(normal, 1e-4f)), reflect3(inc, normal), n1, rw, tc,
                                           r.dist_in_medium, last.color, last.color[3] * 5.0f, r.depth);
            }
            if (refract_i > 0.001f && sp < MAX_RAYS && !has_tir) {
                stack[sp++] = make_ray(sub3(hp, scale3(normal, 1e-4f)), refr_dir, n2, r.weight * refract_i, tc,
                                       0.0f, hv.color, hv.color[3] * 5.0f, r.depth);
            }
        } else { /* opaque, comp:573-618 */
            float emission = hv.props[1] * 10.0f;
            if (emission > 0.0f && r.depth == 0) {
                for (int k = 0; k < 3; k++) fc[k] = fc[k] + tc[k] * sc[k] * emission * r.weight;
                continue;
            } else if (emission > 0.0f) {
                for (int k = 0; k < 3; k++) fc[k] = fc[k] + tc[k] * sc[k] * emission * r.weight / kPI;
                continue;
            }
            if (r.depth == 0) {
                int lit = 1;
                if (mode != O_MODE_PRIMARY) lit = not_in_shadow(c, add3(hp, scale3(normal, 2e-3f)), light);
                for (int k = 0; k < 3; k++) {
                    float direct = gl[k] * (float)lit * ndotl;
                    fc[k] = fc[k] + direct * sc[k] * tc[k] * r.weight / kPI;
                }
            } else {
                float amb = fmax_c(1.0f - o_det_expf(-r.dist_in_medium / 512.0f), 0.01f);
                for (int k = 0; k < 3; k++) fc[k] = fc[k] + amb * sc[k] * tc[k] * r.weight / kPI;
                continue;
            }
            if (mode == O_MODE_FULL) {
                for (int i = 0; i < INDIRECT_SAMPLES && sp < MAX_RAYS && r.depth <= BOUNCES; i++) {
                    float rx = rand_f(c), ry = rand_f(c);
                    v3 bd = cosine_hemisphere(normal, rx, ry);
                    float nw = r.weight / (float)INDIRECT_SAMPLES;
                    float tint[4] = {tc[0] * sc[0], tc[1] * sc[1], tc[2] * sc[2], tc[3] * sc[3]};
                    stack[sp++] = make_ray(add3(hp, scale3(normal, 1e-1f)), bd, n1, nw, tint, 0.0f,
                                           last.color, last.color[3] * 5.0f, r.depth + 1);
                }
            }
        }
    }
    out_rgb[0] = fc[0]; out_rgb[1] = fc[1]; out_rgb[2] = fc[2];
}

static inline uint8_t unorm8(float v) { /* C7 */
    float c = fmin_c(fmax_c(v, 0.0f), 1.0f);
    return (uint8_t)rintf(c * 255.0f);
}

static inline void mat_vec(const float *m, float x, float y, float z, float w, float out[4]) { /* C1 */
    for (int r = 0; r < 4; r++) out[r] = (m[0 * 4 + r] * x + m[1 * 4 + r] * y) + (m[2 * 4 + r] * z + m[3 * 4 + r] * w);
}

void o_scene_defaults(o_scene *s) {
    memset(s, 0, sizeof *s);
    s->tex_dim = 1;
    s->voxel_scale = 1.0f;
    for (int i = 0; i < 3; i++) { s->bounds_min[i] = -1023; s->bounds_max[i] = 1024; s->highlighted[i] = -1; }
    s->global_light[0] = s->global_light[1] = s->global_light[2] = s->global_light[3] = 1.0f;
    /* glm::normalize(vec3(0.3481553, 0.870388, 0.3481553)), main.cpp:483 */
    float l[3] = {0.3481553f, 0.870388f, 0.3481553f};
    float t0 = l[0] * l[0], t1 = l[1] * l[1], t2 = l[2] * l[2];
    float inv = 1.0f / sqrtf(t0 + t1 + t2);
    s->light_dir[0] = l[0] * inv; s->light_dir[1] = l[1] * inv; s->light_dir[2] = l[2] * inv;
}

/* comp:624-645 main, one call per row range */
void o_render(const o_scene *s, int W, int H, int row0, int row1, int mode, uint8_t *rgba8, int32_t *id_dist,
              uint32_t *fetch_map, o_stats *stats) {
    ctx_t c;
    memset(&c, 0, sizeof c);
    c.s = s;
    for (int py = row0; py < row1; py++) {
        for (int px = 0; px < W; px++) {
            c.px_fetches = 0;
            c.px_index = (uint32_t)(py * W + px);
            init_rng(&c, px, py, 0);
            float u = ((float)px / (float)W) * 2.0f - 1.0f;
            float v = ((float)py / (float)H) * 2.0f - 1.0f;
            float view[4];
            mat_vec(s->inv_proj, u, v, -1.0f, 1.0f, view);
            if (fabsf(view[3]) > 1e-6f) { float w = view[3]; view[0] /= w; view[1] /= w; view[2] /= w; view[3] /= w; }
            v3 vd = {view[0], view[1], view[2]};
            vd = normalize3(vd);
            float wd4[4];
            mat_vec(s->inv_view, vd.x, vd.y, vd.z, 0.0f, wd4);
            v3 wd = {wd4[0], wd4[1], wd4[2]};
            wd = normalize3(wd);
            v3 ro = {s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]};
            float rgb[3];
            int32_t vid, dist;
            path_trace(&c, ro, wd, mode, rgb, &vid, &dist);
            size_t p = (size_t)py * (size_t)W + (size_t)px;
            if (rgba8) { rgba8[p * 4 + 0] = unorm8(rgb[0]); rgba8[p * 4 + 1] = unorm8(rgb[1]); rgba8[p * 4 + 2] = unorm8(rgb[2]); rgba8[p * 4 + 3] = 255; }
            if (id_dist) { id_dist[p * 2 + 0] = vid; id_dist[p * 2 + 1] = dist; }
            if (fetch_map) fetch_map[p] = c.px_fetches;
        }
    }
    if (stats) {
        stats->fetches += c.st.fetches; stats->finds += c.st.finds; stats->root_restarts += c.st.root_restarts;
        stats->steps += c.st.steps; stats->hits += c.st.hits; stats->shadow_rays += c.st.shadow_rays;
    }
}

/* Independent point query (no parent cache, fresh descent) for find cross-checks */
int o_find_point(const o_scene *s, const int32_t pos[3], uint8_t leaf[8], int32_t mn[3], int32_t mx[3]) {
    ctx_t c;
    memset(&c, 0, sizeof c);
    c.s = s;
    i3 p = {pos[0], pos[1], pos[2]};
    i3 nmin = {s->bounds_min[0], s->bounds_min[1], s->bounds_min[2]};
    i3 nmax = {s->bounds_max[0], s->bounds_max[1], s->bounds_max[2]};
    int32_t cur = 0;
    /* walk manually so the raw leaf texels can be returned */
    int32_t coord = 0;
    memset(leaf, 0, 8);
    if (!in_world(s, p)) { mn[0] = nmin.x; mn[1] = nmin.y; mn[2] = nmin.z; mx[0] = nmax.x; mx[1] = nmax.y; mx[2] = nmax.z; return 0; }
    (void)cur;
    for (int i = 0; i < 16; i++) {
        uint32_t nd = fetch(&c, coord);
        uint32_t base = s->wide_pointers ? (uint32_t)coord + 1u : (nd & 0x7fffffu), mask = nd >> 24;
        i3 mid = {nmin.x + (nmax.x - nmin.x) / 2, nmin.y + (nmax.y - nmin.y) / 2, nmin.z + (nmax.z - nmin.z) / 2};
        int ci = (p.x >= mid.x ? 4 : 0) + (p.y >= mid.y ? 2 : 0) + (p.z >= mid.z ? 1 : 0);
        if (ci & 4) nmin.x = mid.x; else nmax.x = mid.x;
        if (ci & 2) nmin.y = mid.y; else nmax.y = mid.y;
        if (ci & 1) nmin.z = mid.z; else nmax.z = mid.z;
        mn[0] = nmin.x; mn[1] = nmin.y; mn[2] = nmin.z; mx[0] = nmax.x; mx[1] = nmax.y; mx[2] = nmax.z;
        if (!((mask >> ci) & 1u)) return 0;
        uint32_t ptr = fetch(&c, (int32_t)(base + (uint32_t)popc8(mask & ((1u << ci) - 1u))));
        coord = (int32_t)((ptr & 0x7fffffu) | (s->wide_pointers ? (ptr >> 24) << 23 : 0u));
        if (ptr & 0x800000u) {
            uint32_t a = fetch(&c, coord), b = fetch(&c, coord + 1);
            memcpy(leaf, &a, 4); memcpy(leaf + 4, &b, 4);
            return 1;
        }
    }
    return 0;
}

/* ---- display pass: ID-aware box blur (shaders/quad.frag:22-83) ------------------------------------
 * Conventions: sampler2D on an rgba8 image returns byte/255.0f; the 8-bit framebuffer write is the C7
 * rounding; sums run in the shader's loop order (y outer, x inner), in fp32. */
void o_denoise(const uint8_t *rgba8, const int32_t *id_dist, int W, int H, uint8_t *out) {
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            size_t p = (size_t)py * W + px;
            int center_id = id_dist[p * 2], center_dist = id_dist[p * 2 + 1];
            if (center_id == 0) { memcpy(out + p * 4, rgba8 + p * 4, 4); continue; }          /* quad.frag:36-39 */
            float radius_f = 200.0f / sqrtf((float)(center_dist > 1 ? center_dist : 1));       /* :45 */
            int R = (int)radius_f;
            R = R < 1 ? 1 : (R > 20 ? 20 : R);                                                    /* :48 */
            float sum[3] = {0.0f, 0.0f, 0.0f}, count = 0.0f;
            for (int y = -R; y <= R; y++)
                for (int x = -R; x <= R; x++) {
                    int nx = px + x, ny = py + y;
                    if (nx < 0 || nx >= W || ny < 0 || ny >= H) continue;                         /* :60-63 */
                    size_t q = (size_t)ny * W + nx;
                    if (id_dist[q * 2] == center_id) {                                            /* :67-73 */
                        sum[0] = sum[0] + (float)rgba8[q * 4 + 0] / 255.0f;
                        sum[1] = sum[1] + (float)rgba8[q * 4 + 1] / 255.0f;
                        sum[2] = sum[2] + (float)rgba8[q * 4 + 2] / 255.0f;
                        count = count + 1.0f;
                    }
                }
            float d = fmax_c(count, 1.0f);                                                        /* :78 */
            out[p * 4 + 0] = unorm8(sum[0] / d);
            out[p * 4 + 1] = unorm8(sum[1] / d);
            out[p * 4 + 2] = unorm8(sum[2] / d);
            out[p * 4 + 3] = 255;
        }
}
