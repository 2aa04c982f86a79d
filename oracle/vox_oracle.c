/*
 * vox_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of the reference's MagicaVoxel loader, src/voxReader.cpp.
 * The file is read through a tiny in-memory FILE emulation so that the
 * reference's unchecked fread/ftell/fseek sequencing is reproduced as written.
 */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SAFE_MIN (-2048) /* voxReader.cpp:18 */
#define SAFE_MAX 2048    /* voxReader.cpp:19 */

typedef struct { const uint8_t *p; long len, pos; } mfile;

/* fread semantics: copies what is available in whole items, returns item count */
static size_t m_read(void *dst, size_t sz, size_t n, mfile *f) {
    long avail = f->len - f->pos;
    if (avail < 0) avail = 0;
    size_t items = sz ? (size_t)avail / sz : 0;
    if (items > n) items = n;
    memcpy(dst, f->p + f->pos, items * sz);
    f->pos += (long)(items * sz);
    return items;
}

typedef struct { int32_t sx, sy, sz; size_t n; uint8_t *xyzi; } vmodel;

enum { N_TRN, N_GRP, N_SHP };
typedef struct {
    int id, type;
    int child;            /* TRN */
    float t[3];
    uint8_t rot;
    int *kids; int n_kids; /* GRP */
    int model;            /* SHP */
} snode;

typedef struct { char *k, *v; } kv;
typedef struct { kv *e; int n; } dict;

/* voxReader.cpp:51-59 */
static char *read_string(mfile *f) {
    int32_t size;
    if (m_read(&size, 4, 1, f) != 1) return (char *)calloc(1, 1);
    if (size <= 0 || size > 1024 * 1024) return (char *)calloc(1, 1);
    char *s = (char *)calloc((size_t)size + 1, 1);
    if (m_read(s, 1, (size_t)size, f) != (size_t)size) { free(s); return (char *)calloc(1, 1); }
    /* std::string(buffer.data()) stops at the first NUL */
    return s;
}

/* voxReader.cpp:61-72 (std::map: later duplicates overwrite) */
static dict read_dict(mfile *f) {
    dict d = {NULL, 0};
    int32_t pairs;
    if (m_read(&pairs, 4, 1, f) != 1) return d;
    if (pairs < 0 || pairs > 1000) return d;
    d.e = (kv *)calloc((size_t)pairs + 1, sizeof(kv));
    for (int i = 0; i < pairs; i++) {
        char *k = read_string(f), *v = read_string(f);
        int j;
        for (j = 0; j < d.n; j++) if (!strcmp(d.e[j].k, k)) break;
        if (j < d.n) { free(d.e[j].v); d.e[j].v = v; free(k); }
        else { d.e[d.n].k = k; d.e[d.n].v = v; d.n++; }
    }
    return d;
}
static const char *dict_get(const dict *d, const char *k) {
    for (int i = 0; i < d->n; i++) if (!strcmp(d->e[i].k, k)) return d->e[i].v;
    return NULL;
}
static void dict_free(dict *d) {
    for (int i = 0; i < d->n; i++) { free(d->e[i].k); free(d->e[i].v); }
    free(d->e);
}

/* voxReader.cpp:75-81 */
static int safe_round(float v) { return v >= 0.0f ? (int)(v + 0.5f) : (int)(v - 0.5f); }

/* column-major 4x4, m[c*4+r] like glm */
static void mat_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }
/* glm/detail/type_mat4x4.inl:630-648 : R[c] = ((A0*B[c][0] + A1*B[c][1]) + A2*B[c][2]) + A3*B[c][3] */
static void mat_mul(const float *a, const float *b, float *r) {
    float out[16];
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = a[0 * 4 + row] * b[c * 4 + 0];
            s = s + a[1 * 4 + row] * b[c * 4 + 1];
            s = s + a[2 * 4 + row] * b[c * 4 + 2];
            s = s + a[3 * 4 + row] * b[c * 4 + 3];
            out[c * 4 + row] = s;
        }
    memcpy(r, out, 64);
}

/* voxReader.cpp:84-117 */
static void rotation_matrix(uint8_t rb, float *m) {
    int r0 = rb & 3, r1 = (rb >> 2) & 3;
    float s0 = (rb & 16) ? -1.0f : 1.0f, s1 = (rb & 32) ? -1.0f : 1.0f;
    int s2neg = (rb & 64) != 0;
    float row0[3] = {0, 0, 0}, row1[3] = {0, 0, 0}, row2[3];
    /* r0/r1 may be 3 in malformed files: glm::vec3 operator[] with index 3 is
     * out of range upstream; treated here as a no-op (never produced by MagicaVoxel). */
    if (r0 < 3) row0[r0] = s0;
    if (r1 < 3) row1[r1] = s1;
    row2[0] = row0[1] * row1[2] - row1[1] * row0[2]; /* glm::cross */
    row2[1] = row0[2] * row1[0] - row1[2] * row0[0];
    row2[2] = row0[0] * row1[1] - row1[0] * row0[1];
    if (s2neg) { row2[0] = -row2[0]; row2[1] = -row2[1]; row2[2] = -row2[2]; }
    mat_identity(m);
    m[0 * 4 + 0] = row0[0]; m[1 * 4 + 0] = row0[1]; m[2 * 4 + 0] = row0[2];
    m[0 * 4 + 1] = row1[0]; m[1 * 4 + 1] = row1[1]; m[2 * 4 + 1] = row1[2];
    m[0 * 4 + 2] = row2[0]; m[1 * 4 + 2] = row2[1]; m[2 * 4 + 2] = row2[2];
}

typedef struct {
    snode *nodes; int n_nodes;
    vmodel *models; int n_models;
    uint32_t palette[256];
    o_octree *tree;
    int origin[3];
    long inserted;
} gctx;

static snode *find_node(gctx *g, int id) {
    /* std::map keyed by id: last write wins -> search from the end */
    for (int i = g->n_nodes - 1; i >= 0; i--) if (g->nodes[i].id == id) return &g->nodes[i];
    return NULL;
}

static const o_voxel k_default_voxel = {3.0f, 0.0f, 0.0f}; /* voxReader.cpp:21 -> main.cpp:220-221 voxels[0] */

/* voxReader.cpp:121-211 */
static void traverse(gctx *g, int id, const float *parent) {
    snode *n = find_node(g, id);
    if (!n) return;
    if (n->type == N_TRN) {
        float tr[16], rot[16], tmp[16], cur[16];
        mat_identity(tr); /* glm::translate(mat4(1), t): col3 = m0*t0 + m1*t1 + m2*t2 + m3 */
        tr[12] = 1.0f * n->t[0] + 0.0f * n->t[1] + 0.0f * n->t[2] + 0.0f;
        tr[13] = 0.0f * n->t[0] + 1.0f * n->t[1] + 0.0f * n->t[2] + 0.0f;
        tr[14] = 0.0f * n->t[0] + 0.0f * n->t[1] + 1.0f * n->t[2] + 0.0f;
        tr[15] = 1.0f;
        rotation_matrix(n->rot, rot);
        mat_mul(parent, tr, tmp);
        mat_mul(tmp, rot, cur);
        traverse(g, n->child, cur);
    } else if (n->type == N_GRP) {
        for (int i = 0; i < n->n_kids; i++) traverse(g, n->kids[i], parent);
    } else {
        if (n->model < 0 || n->model >= g->n_models) return;
        const vmodel *m = &g->models[n->model];
        float cx = (float)m->sx / 2.0f, cy = (float)m->sy / 2.0f, cz = (float)m->sz / 2.0f;
        for (size_t i = 0; i < m->n; i++) {
            const uint8_t *v = &m->xyzi[i * 4];
            int ci = (int)v[3] - 1;
            if (ci < 0 || ci >= 256) ci = 0;
            float lx = (float)v[0] - cx, ly = (float)v[1] - cy, lz = (float)v[2] - cz, lw = 1.0f;
            /* glm mat4*vec4: (m0*x + m1*y) + (m2*z + m3*w) */
            float f[3];
            for (int r = 0; r < 3; r++)
                f[r] = (parent[0 * 4 + r] * lx + parent[1 * 4 + r] * ly) +
                       (parent[2 * 4 + r] * lz + parent[3 * 4 + r] * lw);
            int fx = g->origin[0] + safe_round(f[0]);
            int fy = g->origin[1] + safe_round(f[2]);
            int fz = g->origin[2] + safe_round(f[1]);
            if (fx < SAFE_MIN || fx > SAFE_MAX || fy < SAFE_MIN || fy > SAFE_MAX ||
                fz < SAFE_MIN || fz > SAFE_MAX) continue;
            o_voxel_obj vo;
            vo.voxel = k_default_voxel;
            vo.color = g->palette[ci];
            vo.coord.x = fx; vo.coord.y = fy; vo.coord.z = fz;
            o_octree_insert(g->tree, vo);
            g->inserted++;
        }
    }
}

/* voxReader.cpp:215-418 */
int o_load_vox_mem(const uint8_t *buf, size_t len, o_octree *tree, int ox, int oy, int oz, long *n_inserted) {
    if (n_inserted) *n_inserted = 0;
    if (!tree) return 0;
    mfile f = {buf, (long)len, 0};
    char hdr[4];
    int32_t version;
    if (m_read(hdr, 1, 4, &f) != 4) return 0;
    if (m_read(&version, 4, 1, &f) != 1) return 0;
    if (strncmp(hdr, "VOX ", 4) != 0) return 0;

    gctx g;
    memset(&g, 0, sizeof g);
    g.tree = tree;
    g.origin[0] = ox; g.origin[1] = oy; g.origin[2] = oz;
    for (int i = 0; i < 256; i++) /* default grayscale palette :244-246; make_color_rgba color.c:9-12 */
        g.palette[i] = ((uint32_t)i << 24) | ((uint32_t)i << 16) | ((uint32_t)i << 8) | 255u;
    int32_t last[3] = {0, 0, 0};
    long file_size = (long)len;
    f.pos = 8;

    while (f.pos < file_size - 12) {
        char id[4];
        int32_t content, children;
        if (m_read(id, 1, 4, &f) < 4) break;
        if (m_read(&content, 4, 1, &f) < 1) break;
        if (m_read(&children, 4, 1, &f) < 1) break;
        if (content < 0 || children < 0) break;
        long next = f.pos + content, end = next + children;
        if (end > file_size) break;

        if (!strncmp(id, "MAIN", 4)) continue;
        else if (!strncmp(id, "PACK", 4)) f.pos += content;
        else if (!strncmp(id, "SIZE", 4)) {
            m_read(&last[0], 4, 1, &f); m_read(&last[1], 4, 1, &f); m_read(&last[2], 4, 1, &f);
        } else if (!strncmp(id, "XYZI", 4)) {
            int32_t nv = 0;
            m_read(&nv, 4, 1, &f);
            if (nv < 0 || nv > 10000000) { f.pos = end; continue; }
            g.models = (vmodel *)realloc(g.models, sizeof(vmodel) * (size_t)(g.n_models + 1));
            vmodel *m = &g.models[g.n_models++];
            m->sx = last[0]; m->sy = last[1]; m->sz = last[2];
            m->n = (size_t)nv;
            m->xyzi = (uint8_t *)calloc((size_t)nv + 1, 4);
            for (int i = 0; i < nv; i++) {
                m_read(&m->xyzi[i * 4 + 0], 1, 1, &f); m_read(&m->xyzi[i * 4 + 1], 1, 1, &f);
                m_read(&m->xyzi[i * 4 + 2], 1, 1, &f); m_read(&m->xyzi[i * 4 + 3], 1, 1, &f);
            }
        } else if (!strncmp(id, "RGBA", 4)) {
            for (int i = 0; i < 256; i++) {
                uint8_t c[4] = {0, 0, 0, 0};
                m_read(&c[0], 1, 1, &f); m_read(&c[1], 1, 1, &f); m_read(&c[2], 1, 1, &f); m_read(&c[3], 1, 1, &f);
                g.palette[i] = ((uint32_t)c[0] << 24) | ((uint32_t)c[1] << 16) | ((uint32_t)c[2] << 8) | c[3];
            }
        } else if (!strncmp(id, "nTRN", 4) || !strncmp(id, "nGRP", 4) || !strncmp(id, "nSHP", 4)) {
            g.nodes = (snode *)realloc(g.nodes, sizeof(snode) * (size_t)(g.n_nodes + 1));
            snode *n = &g.nodes[g.n_nodes++];
            memset(n, 0, sizeof *n);
            n->child = -1; n->model = -1; n->rot = 4;
            m_read(&n->id, 4, 1, &f);
            dict attrs = read_dict(&f);
            dict_free(&attrs);
            if (id[1] == 'T') {
                n->type = N_TRN;
                int32_t reserved, layer, frames = 0;
                m_read(&n->child, 4, 1, &f);
                m_read(&reserved, 4, 1, &f); m_read(&layer, 4, 1, &f); m_read(&frames, 4, 1, &f);
                for (int i = 0; i < frames; i++) {
                    dict d = read_dict(&f);
                    if (i == 0) {
                        const char *t = dict_get(&d, "_t"), *r = dict_get(&d, "_r");
                        if (t) { /* stringstream >> float x3: stops at the first failed extraction */
                            char *e1, *e2, *e3;
                            float a = strtof(t, &e1);
                            if (e1 != t) { n->t[0] = a; float b = strtof(e1, &e2);
                                if (e2 != e1) { n->t[1] = b; float c = strtof(e2, &e3);
                                    if (e3 != e2) n->t[2] = c; } }
                        }
                        if (r) n->rot = (uint8_t)atoi(r);
                    }
                    dict_free(&d);
                }
            } else if (id[1] == 'G') {
                n->type = N_GRP;
                int32_t nk = 0;
                m_read(&nk, 4, 1, &f);
                if (nk < 0) nk = 0;
                n->kids = (int *)calloc((size_t)nk + 1, sizeof(int));
                for (int i = 0; i < nk; i++) {
                    int32_t k = 0;
                    if (m_read(&k, 4, 1, &f) != 1) { nk = i; break; }
                    n->kids[n->n_kids++] = k;
                }
            } else {
                n->type = N_SHP;
                int32_t nm = 0;
                m_read(&nm, 4, 1, &f);
                for (int i = 0; i < nm && f.pos < f.len; i++) {
                    int32_t mid = 0;
                    m_read(&mid, 4, 1, &f);
                    dict d = read_dict(&f);
                    dict_free(&d);
                    if (i == 0) n->model = mid;
                }
            }
        }
        f.pos = end;
    }

    int ret;
    if (g.n_nodes == 0) { /* RAW mode :382-408 */
        long count = 0;
        for (int mi = 0; mi < g.n_models; mi++) {
            const vmodel *m = &g.models[mi];
            for (size_t i = 0; i < m->n; i++) {
                const uint8_t *v = &m->xyzi[i * 4];
                int ci = (int)v[3] - 1;
                if (ci < 0 || ci >= 256) ci = 0;
                int fx = ox + v[0], fy = oy + v[2], fz = oz + v[1];
                if (fx < SAFE_MIN || fx > SAFE_MAX || fy < SAFE_MIN || fy > SAFE_MAX ||
                    fz < SAFE_MIN || fz > SAFE_MAX) continue;
                o_voxel_obj vo;
                vo.voxel = k_default_voxel;
                vo.color = g.palette[ci];
                vo.coord.x = fx; vo.coord.y = fy; vo.coord.z = fz;
                o_octree_insert(tree, vo);
                count++;
            }
        }
        g.inserted = count;
        ret = count > 0;
    } else { /* scene graph :411-417 */
        if (find_node(&g, 0)) {
            float I[16];
            mat_identity(I);
            traverse(&g, 0, I);
        }
        ret = 1;
    }
    if (n_inserted) *n_inserted = g.inserted;
    for (int i = 0; i < g.n_models; i++) free(g.models[i].xyzi);
    free(g.models);
    for (int i = 0; i < g.n_nodes; i++) free(g.nodes[i].kids);
    free(g.nodes);
    return ret;
}

int o_load_vox_file(const char *path, o_octree *tree, int ox, int oy, int oz, long *n_inserted) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return 0;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)n + 1);
    size_t got = fread(buf, 1, (size_t)n, fp);
    fclose(fp);
    int r = o_load_vox_mem(buf, got, tree, ox, oy, oz, n_inserted);
    free(buf);
    return r;
}
