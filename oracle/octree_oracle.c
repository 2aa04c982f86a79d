/*
 * octree_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's pointer octree: build, merge, remove,
 * flatten into the RGBA8UI texel stream, and the CPU ray cast.
 * Follows src/octree.cpp; line numbers cited per function.
 *
 * Third-party arithmetic restated here: lib/libvmm.a (binary only, no source,
 * version unknown). Semantics taken from SURVEY.md 8(c) (disassembly):
 *   ivec3_add/sub component-wise; ivec3_scalar_div truncating, /0 -> 0;
 *   ivec3_vec3 truncates toward zero;
 *   ivec3_equal_vec(a,b) = a.x==b.x && a.y!=0 && b.y!=0 && a.z==b.z  (sic).
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define O_MIN_HEIGHT (-1024) /* octree.cpp:12-14 */

/* ---- vmm restatements ---------------------------------------------------- */
static o_ivec3 iv_add(o_ivec3 a, o_ivec3 b) { o_ivec3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static o_ivec3 iv_sub(o_ivec3 a, o_ivec3 b) { o_ivec3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static o_ivec3 iv_sdiv(o_ivec3 a, int s) {
    o_ivec3 r = {0, 0, 0};
    if (s != 0) { r.x = a.x / s; r.y = a.y / s; r.z = a.z / s; }
    return r;
}
static o_ivec3 iv_from_vec3(o_vec3 v) { o_ivec3 r = {(int32_t)v.x, (int32_t)v.y, (int32_t)v.z}; return r; }
/* the shipped library's "equality": y is only tested for non-zero */
static int iv_equal_quirky(o_ivec3 a, o_ivec3 b) { return a.x == b.x && a.y != 0 && b.y != 0 && a.z == b.z; }

/* octree.cpp:36-40 */
static o_voxel_obj invalid_voxel(void) {
    o_voxel_obj v;
    memset(&v, 0, sizeof v);
    v.coord.y = O_MIN_HEIGHT;
    return v;
}

/* octree.cpp:46-76 : x>=mid -> bit2, y>=mid -> bit1, z>=mid -> bit0 */
static int child_slot(o_ivec3 c, o_ivec3 mid) {
    return (c.x >= mid.x ? 4 : 0) | (c.y >= mid.y ? 2 : 0) | (c.z >= mid.z ? 1 : 0);
}

/* octree.cpp:80-87 */
static int outside(o_ivec3 c, o_ivec3 lo, o_ivec3 hi) {
    return c.x < lo.x || c.x >= hi.x || c.y < lo.y || c.y >= hi.y || c.z < lo.z || c.z >= hi.z;
}

/* octree.cpp:89-100 */
o_octree *o_octree_create(o_octree *parent, o_ivec3 lbb, o_ivec3 rtf) {
    o_octree *n = (o_octree *)calloc(1, sizeof *n);
    n->parent = parent;
    n->lbb = lbb;
    n->rtf = rtf;
    return n;
}

/* octree.cpp:102-130 */
o_voxel_obj o_octree_find(o_octree *t, o_ivec3 c) {
    if (outside(c, t->lbb, t->rtf)) return invalid_voxel();
    if (t->has_voxel && iv_equal_quirky(t->voxel.coord, c)) return t->voxel;
    if (!t->children) return invalid_voxel();
    o_ivec3 mid = iv_sdiv(iv_add(t->lbb, t->rtf), 2); /* note: (lo+hi)/2, not lo+(hi-lo)/2 */
    o_octree *r = t->children[child_slot(c, mid)];
    for (;;) {
        if (!r) return invalid_voxel();
        if (outside(c, r->lbb, r->rtf)) return invalid_voxel();
        if (r->has_voxel && iv_equal_quirky(r->voxel.coord, c)) return r->voxel;
        if (!r->children) return invalid_voxel();
        mid = iv_sdiv(iv_add(r->lbb, r->rtf), 2);
        r = r->children[child_slot(c, mid)];
    }
}

/* octree.cpp:132-182 */
static int create_children(o_octree *t) {
    t->children = (o_octree **)malloc(sizeof(o_octree *) * 8);
    if (!t->children) return -1;
    o_ivec3 lo = t->lbb, hi = t->rtf;
    o_ivec3 sz = iv_sub(hi, lo);
    if (sz.x <= 1 && sz.y <= 1 && sz.z <= 1) return 0;
    o_ivec3 mid = {lo.x + (hi.x - lo.x) / 2, lo.y + (hi.y - lo.y) / 2, lo.z + (hi.z - lo.z) / 2};
    for (int i = 0; i < 8; i++) {
        o_ivec3 a, b;
        a.x = (i & 4) ? mid.x : lo.x; b.x = (i & 4) ? hi.x : mid.x;
        a.y = (i & 2) ? mid.y : lo.y; b.y = (i & 2) ? hi.y : mid.y;
        a.z = (i & 1) ? mid.z : lo.z; b.z = (i & 1) ? hi.z : mid.z;
        t->children[i] = o_octree_create(t, a, b);
        t->children[i]->voxel = invalid_voxel();
    }
    /* :174-179 -- one child unconditionally inherits the parent's voxel and is
     * flagged has_voxel even when the parent held nothing. */
    int pos = child_slot(t->voxel.coord, mid);
    t->children[pos]->voxel = t->voxel;
    t->children[pos]->has_voxel = 1;
    t->voxel = invalid_voxel();
    t->has_voxel = 0;
    return 0;
}

/* octree.cpp:185-200 */
static int is_leaf(o_octree *n) { return n && n->has_voxel && !n->children; }
static int identical(o_octree *a, o_octree *b) {
    if (!is_leaf(a) || !is_leaf(b)) return 0;
    return (a->voxel.color == b->voxel.color &&
            a->voxel.voxel.refraction == b->voxel.voxel.refraction &&
            a->voxel.voxel.illumination == b->voxel.voxel.illumination) ||
           (a->voxel.coord.y <= O_MIN_HEIGHT && b->voxel.coord.y <= O_MIN_HEIGHT);
}

/* octree.cpp:203-255 */
static int split_node(o_octree *t) {
    if (t->children) return 0;
    o_voxel_obj orig = t->voxel;
    int was_solid = t->has_voxel;
    o_ivec3 lo = t->lbb, hi = t->rtf;
    o_ivec3 mid = {lo.x + (hi.x - lo.x) / 2, lo.y + (hi.y - lo.y) / 2, lo.z + (hi.z - lo.z) / 2};
    if (create_children(t) != 0) return -1;
    if (was_solid) {
        if (iv_equal_quirky(orig.coord, t->lbb)) { /* merged volume: fill all 8 */
            for (int i = 0; i < 8; i++) {
                t->children[i]->voxel = orig;
                t->children[i]->has_voxel = 1;
                t->children[i]->voxel.coord = t->children[i]->lbb;
            }
        } else { /* lazy point: move into its octant */
            int pos = child_slot(orig.coord, mid);
            t->children[pos]->voxel = orig;
            t->children[pos]->has_voxel = 1;
        }
        t->has_voxel = 0;
    }
    return 0;
}

/* octree.cpp:258-285 */
static void try_merge(o_octree *n) {
    if (!n || !n->children) return;
    for (int i = 0; i < 8; i++) if (!is_leaf(n->children[i])) return;
    o_octree *first = n->children[0];
    for (int i = 1; i < 8; i++) if (!identical(first, n->children[i])) return;
    n->voxel = first->voxel;
    n->voxel.coord = n->lbb;
    n->has_voxel = 1;
    for (int i = 0; i < 8; i++) free(n->children[i]);
    free(n->children);
    n->children = NULL;
}

/* octree.cpp:287-323 */
void o_octree_insert(o_octree *t, o_voxel_obj v) {
    if (!t) return;
    if (outside(v.coord, t->lbb, t->rtf)) return;
    o_ivec3 sz = iv_sub(t->rtf, t->lbb);
    if (sz.x <= 1 && sz.y <= 1 && sz.z <= 1) {
        t->voxel = v;
        t->has_voxel = 1;
        return;
    }
    if (!t->children && split_node(t) != 0) return;
    o_ivec3 mid = {t->lbb.x + sz.x / 2, t->lbb.y + sz.y / 2, t->lbb.z + sz.z / 2};
    o_octree_insert(t->children[child_slot(v.coord, mid)], v);
    try_merge(t);
}

/* octree.cpp:684-740 */
void o_octree_remove(o_octree *t, o_ivec3 c) {
    if (!t) return;
    if (outside(c, t->lbb, t->rtf)) return;
    o_ivec3 sz = iv_sub(t->rtf, t->lbb);
    if (sz.x <= 1 && sz.y <= 1 && sz.z <= 1) { t->has_voxel = 0; return; }
    if (!t->children && t->has_voxel && split_node(t) != 0) return;
    if (!t->children) return;
    o_ivec3 mid = {t->lbb.x + sz.x / 2, t->lbb.y + sz.y / 2, t->lbb.z + sz.z / 2};
    o_octree_remove(t->children[child_slot(c, mid)], c);
    int all_empty = 1;
    for (int i = 0; i < 8; i++)
        if (t->children[i]->has_voxel || t->children[i]->children) { all_empty = 0; break; }
    if (all_empty) {
        for (int i = 0; i < 8; i++) free(t->children[i]);
        free(t->children);
        t->children = NULL;
        t->has_voxel = 0;
    }
}

/* octree.cpp:743-754 */
void o_octree_delete(o_octree *t) {
    if (!t) return;
    if (t->children) {
        for (int i = 0; i < 8; i++) o_octree_delete(t->children[i]);
        free(t->children);
    }
    free(t);
}

/* ---- CPU ray cast (config 1) -------------------------------------------- */
/* octree.cpp:364-403 */
static o_octree *find_leaf(o_octree *root, o_ivec3 p, o_ivec3 *nmin, o_ivec3 *nmax) {
    o_octree *cur = root;
    o_ivec3 lo = root->lbb, hi = root->rtf;
    if (outside(p, lo, hi)) return NULL;
    while (cur->children) {
        o_ivec3 mid = {lo.x + (hi.x - lo.x) / 2, lo.y + (hi.y - lo.y) / 2, lo.z + (hi.z - lo.z) / 2};
        int ci = child_slot(p, mid);
        if (ci & 4) lo.x = mid.x; else hi.x = mid.x;
        if (ci & 2) lo.y = mid.y; else hi.y = mid.y;
        if (ci & 1) lo.z = mid.z; else hi.z = mid.z;
        cur = cur->children[ci];
        if (!cur) { *nmin = lo; *nmax = hi; return NULL; }
    }
    *nmin = lo; *nmax = hi;
    return cur;
}

/* octree.cpp:405-485 */
o_octree *o_octree_ray_cast(o_octree *root, o_vec3 origin, o_vec3 dir, o_vec3 wmin, o_vec3 wmax) {
    o_vec3 rp = origin, rd = dir, inv;
    inv.x = (fabsf(rd.x) < 1e-8f) ? 1e20f : 1.0f / rd.x;
    inv.y = (fabsf(rd.y) < 1e-8f) ? 1e20f : 1.0f / rd.y;
    inv.z = (fabsf(rd.z) < 1e-8f) ? 1e20f : 1.0f / rd.z;
    o_ivec3 mp = {(int)floorf(rp.x), (int)floorf(rp.y), (int)floorf(rp.z)};
    o_ivec3 nmin = iv_from_vec3(wmin), nmax = iv_from_vec3(wmax);
    for (int i = 0; i < 512; i++) {
        o_octree *n = find_leaf(root, mp, &nmin, &nmax);
        if (n && n->has_voxel && n->voxel.coord.y > O_MIN_HEIGHT) return n;
        float tx = (rd.x > 0.0f ? (float)nmax.x - rp.x : (float)nmin.x - rp.x) * inv.x;
        float ty = (rd.y > 0.0f ? (float)nmax.y - rp.y : (float)nmin.y - rp.y) * inv.y;
        float tz = (rd.z > 0.0f ? (float)nmax.z - rp.z : (float)nmin.z - rp.z) * inv.z;
        float m_yz = ty < tz ? ty : tz;
        float t = tx < m_yz ? tx : m_yz;
        int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
        if (t < 0.0001f) t = 0.0001f;
        rp.x += rd.x * t; rp.y += rd.y * t; rp.z += rd.z * t;
        o_vec3 tp = rp;
        if (axis == 0) tp.x += rd.x * 0.001f;
        else if (axis == 1) tp.y += rd.y * 0.001f;
        else tp.z += rd.z * 0.001f;
        mp.x = (int)floorf(tp.x); mp.y = (int)floorf(tp.y); mp.z = (int)floorf(tp.z);
        if (outside(mp, root->lbb, root->rtf)) return NULL;
    }
    return NULL;
}

/* ---- flatten ------------------------------------------------------------- */
/* octree.cpp:488-498 */
static uint8_t child_mask(o_octree *n) {
    if (!n || !n->children) return 0;
    uint8_t m = 0;
    for (int i = 0; i < 8; i++)
        if (n->children[i] && (n->children[i]->has_voxel || n->children[i]->children)) m |= (uint8_t)(1u << i);
    return m;
}
static int popcount8(uint8_t m) { int c = 0; while (m) { m &= (uint8_t)(m - 1); c++; } return c; }

/* octree.cpp:524-552 */
size_t o_octree_texel_size(o_octree *t) {
    if (!t) return 0;
    if (!t->children) return t->has_voxel ? 2 : 0;
    uint8_t m = child_mask(t);
    if (!m) return 0;
    size_t total = 1 + (size_t)popcount8(m);
    for (int i = 0; i < 8; i++) if ((m >> i) & 1) total += o_octree_texel_size(t->children[i]);
    return total;
}

/* octree.cpp:556-570; g_wide (extension, o_octree_texture_wide): address bits 23..30 go to the A byte, which the
 * reference leaves 0 in pointer texels (headers keep their mask there: their list is at header + 1 by construction) */
static int g_wide = 0;
static void encode_pointer(size_t idx, int leaf, uint8_t *out, int is_header) {
    uint32_t v = (uint32_t)idx & 0x7fffffu;
    if (leaf) v |= 0x800000u;
    out[0] = (uint8_t)(v & 0xff);
    out[1] = (uint8_t)((v >> 8) & 0xff);
    out[2] = (uint8_t)((v >> 16) & 0xff);
    if (g_wide && !is_header) out[3] = (uint8_t)((idx >> 23) & 0xff);
}

/* octree.cpp:573-655 ; color getters src/color.c:31-59 */
static void emit(o_octree *n, uint8_t *tex, size_t *next) {
    if (!n) return;
    if (!n->children) {
        if (!n->has_voxel) return;
        size_t b = *next * 4;
        uint32_t c = n->voxel.color;
        tex[b + 0] = (uint8_t)((c >> 24) & 0xff);
        tex[b + 1] = (uint8_t)((c >> 16) & 0xff);
        tex[b + 2] = (uint8_t)((c >> 8) & 0xff);
        tex[b + 3] = 255;
        tex[b + 4] = (uint8_t)(n->voxel.voxel.refraction * 85.0f);
        tex[b + 5] = (uint8_t)(n->voxel.voxel.illumination * 255.0f);
        tex[b + 6] = (uint8_t)(n->voxel.voxel.k * 255.0f);
        tex[b + 7] = (uint8_t)(c & 0xff);
        *next += 2;
        return;
    }
    uint8_t m = child_mask(n);
    if (!m) return;
    size_t header = *next;
    (*next)++;
    size_t ptrs = *next;
    *next += (size_t)popcount8(m);
    encode_pointer(ptrs, 0, &tex[header * 4], 1);
    tex[header * 4 + 3] = m;
    int slot = 0;
    for (int i = 0; i < 8; i++) {
        if (!((m >> i) & 1)) continue;
        size_t child_addr = *next;
        int leaf = (n->children[i]->children == NULL && n->children[i]->has_voxel);
        encode_pointer(child_addr, leaf, &tex[(ptrs + (size_t)slot) * 4], 0);
        emit(n->children[i], tex, next);
        slot++;
    }
}

/* octree.cpp:657-682 */
uint8_t *o_octree_texture(o_octree *t, size_t *arr_size, size_t tex_dim) {
    (void)tex_dim;
    if (!t || !arr_size) return NULL;
    size_t n = o_octree_texel_size(t);
    if (n == 0) { *arr_size = 0; return NULL; }
    *arr_size = n * 4;
    uint8_t *tex = (uint8_t *)calloc(n * 4, 1);
    if (!tex) return NULL;
    size_t next = 0;
    emit(t, tex, &next);
    if (next != n) fprintf(stderr, "oracle: flatten size mismatch %zu vs %zu\n", n, next);
    return tex;
}

/* EXTENSION (oracle.h): the same emit with wide child addresses; not thread-safe (g_wide) */
uint8_t *o_octree_texture_wide(o_octree *t, size_t *arr_size) {
    if (!t || !arr_size) return NULL;
    if (o_octree_texel_size(t) >= ((size_t)1 << 31)) return NULL;
    g_wide = 1;
    uint8_t *tex = o_octree_texture(t, arr_size, 0);
    g_wide = 0;
    return tex;
}

/* src/main.cpp:265-268 */
uint32_t o_tex_dim_for(size_t texels) {
    size_t d = (size_t)ceil(cbrt((double)texels));
    if (d == 0) d = 1;
    return (uint32_t)d;
}

uint64_t o_fnv1a64(const uint8_t *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

/* src/main.cpp:487-503 (the commented-out terrain generator) over a height field given as data, scaled as
 * columns x in [x0, x0+nx), z in [z0, z0+nz); voxels for y in [max(floor_y, h - band), h);
 * loop order and material tests as written there: `if (h == 20 || h == 21) STONE; else if (h == height - 1) DIRT;
 * else GRASS` with 20 -> the column's lower end. Materials voxels[] = {3,0,0} for all three (main.cpp:220-226),
 * colours voxelColors[] GRASS (80,180,60), DIRT (100,70,40), STONE (160,160,160), alpha 255 (main.cpp:247-253);
 * ColorRGBA = R<<24 | G<<16 | B<<8 | A (color.c:9-12). */
int o_fill_heights(o_octree *t, const uint16_t *heights, int size_x, int size_z, int x0, int z0, int nx, int nz,
                   int band, int floor_y) {
    if (!t || !heights || size_x < 1 || size_z < 1 || band < 1 || x0 < 0 || z0 < 0 || nx < 0 || nz < 0 ||
        x0 + nx > size_x || z0 + nz > size_z)
        return -1;
    const uint32_t grass = (80u << 24) | (180u << 16) | (60u << 8) | 255u;
    const uint32_t dirt = (100u << 24) | (70u << 16) | (40u << 8) | 255u;
    const uint32_t stone = (160u << 24) | (160u << 16) | (160u << 8) | 255u;
    for (int i = z0; i < z0 + nz; i++)
        for (int j = x0; j < x0 + nx; j++) {
            const int height = heights[(size_t)i * (size_t)size_x + (size_t)j];
            const int lo = height - band > floor_y ? height - band : floor_y;
            for (int h = lo; h < height; h++) {
                o_voxel_obj v;
                v.coord.x = j; v.coord.y = h; v.coord.z = i;
                v.voxel.refraction = 3.0f; v.voxel.illumination = 0.0f; v.voxel.k = 0.0f;
                if (h == lo || h == lo + 1) v.color = stone;
                else if (h == height - 1) v.color = dirt;
                else v.color = grass;
                o_octree_insert(t, v);
            }
        }
    return 0;
}
