/*
 * oracle.h -- CPU restatement of the reference's hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity checker for the MI355X build. Nothing under
 * oracle/ is part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call it. The product path
 * (voxel-raytracer_amd/csrc) never includes or links anything from here.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the upstream repository root).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - octree build + flatten : pinned by the FNV-1a-64 hashes / texel counts
 *     recorded in SURVEY.md 8(c) for dragon/monu9/nature (tests/golden/flatten.json).
 *   - camera matrices        : pinned by bit patterns produced from the
 *     reference's own Camera.hpp + vendored glm (oracle/_ref, tests/golden/camera.json).
 *   - raytracing.comp restatement : the reference holds no test, golden image
 *     or known-answer vector for the shader and GLSL cannot run in the build
 *     container => "parity unpinned" for the shader arithmetic itself; the
 *     restatement follows the shader line by line in strict IEEE fp32.
 */
#ifndef VRT_ORACLE_H
#define VRT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- basic types (include/vmm/ivec3.h, vec3.h, ray.h) -------- */
typedef struct { int32_t x, y, z; } o_ivec3;
typedef struct { float x, y, z; } o_vec3;

/* include/voxel.hpp:10-18 */
typedef struct { float refraction, illumination, k; } o_voxel;
typedef struct { o_ivec3 coord; uint32_t color; o_voxel voxel; } o_voxel_obj;

/* include/octree.hpp:15-20 */
typedef struct o_octree {
    o_voxel_obj voxel;
    int has_voxel;
    struct o_octree **children, *parent;
    o_ivec3 lbb, rtf;
} o_octree;

/* ---------------- octree (src/octree.cpp) -------------------------------- */
o_octree *o_octree_create(o_octree *parent, o_ivec3 lbb, o_ivec3 rtf); /* :93  */
void o_octree_insert(o_octree *t, o_voxel_obj v);                      /* :287 */
void o_octree_remove(o_octree *t, o_ivec3 coord);                      /* :684 */
o_voxel_obj o_octree_find(o_octree *t, o_ivec3 coord);                 /* :102 */
void o_octree_delete(o_octree *t);                                     /* :743 */
size_t o_octree_texel_size(o_octree *t);                               /* :524 */
/* :657 -- returns calloc'd texel bytes (caller frees), NULL when empty */
uint8_t *o_octree_texture(o_octree *t, size_t *arr_size, size_t tex_dim);
/* :405 -- returns hit node or NULL */
o_octree *o_octree_ray_cast(o_octree *root, o_vec3 origin, o_vec3 dir,
                            o_vec3 world_min, o_vec3 world_max);
/* NON-REFERENCE EXTENSION of :556-570/:657: the same stream with 31-bit child addresses (bits 23..30 in the pointer
 * texel's A byte); see o_scene.wide_pointers. Identical bytes to o_octree_texture() below 2^23 texels. */
uint8_t *o_octree_texture_wide(o_octree *t, size_t *arr_size);
/* tex_dim = ceil(cbrt(texels)), min 1 (src/main.cpp:265-268) */
uint32_t o_tex_dim_for(size_t texels);
/* src/main.cpp:487-503 terrain generator over a height field given as data (config 4) */
int o_fill_heights(o_octree *t, const uint16_t *heights, int size_x, int size_z, int x0, int z0, int nx, int nz,
                   int band, int floor_y);

/* ---------------- .vox loader (src/voxReader.cpp:215-418) ---------------- */
/* Parses a MagicaVoxel file held in memory and inserts into `tree` exactly as
 * load_vox_file does (RAW mode and scene-graph mode). Returns the reference's
 * bool (1/0); *n_inserted receives the number of octree_insert calls. */
int o_load_vox_mem(const uint8_t *buf, size_t len, o_octree *tree,
                   int off_x, int off_y, int off_z, long *n_inserted);
int o_load_vox_file(const char *path, o_octree *tree, int ox, int oy, int oz,
                    long *n_inserted);

/* ---------------- camera (include/Camera.hpp, glm 1.0.0) ------------------ */
typedef struct {
    float position[3], front[3], up[3], right[3], world_up[3];
    float yaw, pitch;
} o_camera;
void o_camera_init(o_camera *c, const float pos[3], float yaw, float pitch); /* Camera.hpp:35-42,86-97 */
void o_camera_view(const o_camera *c, float view[16]);                       /* Camera.hpp:44-47 */
void o_perspective(float fovy_rad, float aspect, float zn, float zf, float m[16]);
void o_mat4_inverse(const float m[16], float inv[16]);
/* src/main.cpp:808-813: fills invProjection, invView, cameraPos (column-major) */
void o_camera_ubo(const o_camera *c, int width, int height,
                  float inv_proj[16], float inv_view[16], float cam_pos[4]);
float o_radians(float deg);

/* ---------------- shader restatement (shaders/raytracing.comp) ------------ */
enum { O_MODE_PRIMARY = 0, O_MODE_PRIMARY_SHADOW = 1, O_MODE_FULL = 2 };

typedef struct {
    /* texture: zero-padded RGBA8UI D^3 volume, linear texel index == (x + D*(y + D*z)) */
    const uint8_t *texels;   /* used bytes (4 per texel) */
    size_t n_texels;         /* texels present; reads beyond return 0 (zero padding) */
    int32_t tex_dim;         /* u_texDim */
    float voxel_scale;       /* u_voxelScale */
    int32_t bounds_min[3];   /* u_worldBoundsMin */
    int32_t bounds_max[3];   /* u_worldBoundsMax */
    float global_light[4];   /* globalLight */
    float light_dir[3];      /* lightDir */
    int32_t highlighted[3];  /* u_highlightedVoxel */
    float inv_proj[16], inv_view[16], cam_pos[4]; /* Camera UBO, column-major */
    /* NON-REFERENCE EXTENSION (0 = the reference's format, the default): a texel stream written by
     * o_octree_texture_wide() for trees beyond the 2^23 texels a 23-bit pointer texel (src/octree.cpp:556-570,
     * comp:89-96) can address. Same texel order and contents; the only differences: a pointer texel's A byte (always
     * 0 in the reference) carries address bits 23..30, and a header's pointer list is taken to start at the header's
     * own address + 1 (where the reference's writer always puts it, octree.cpp:606-625) instead of being read from
     * the header's 23 address bits. For streams below 2^23 texels both readers see the same tree. */
    int32_t wide_pointers;
} o_scene;

typedef struct {
    uint64_t fetches;        /* texelFetch calls issued (every getNodeData) */
    uint64_t finds;          /* octreeFind calls */
    uint64_t root_restarts;  /* finds that restarted from the root */
    uint64_t steps;          /* DDA steps (hitMarching + notInShadow loop bodies) */
    uint64_t hits;           /* primary rays that hit */
    uint64_t shadow_rays;    /* notInShadow invocations */
} o_stats;

void o_scene_defaults(o_scene *s); /* src/main.cpp:478-483,638 */

/* Renders rows [row0,row1) of a W x H frame. Row 0 is v = -1 (bottom).
 * rgba8: W*H*4 bytes, id_dist: W*H*2 int32 (full-frame addressing).
 * fetch_map (optional, W*H uint32): per-pixel texel fetch count. */
void o_render(const o_scene *s, int width, int height, int row0, int row1, int mode,
              uint8_t *rgba8, int32_t *id_dist, uint32_t *fetch_map, o_stats *stats);

/* shaders/quad.frag:22-83 -- the display pass: per-pixel box blur over same-voxelID neighbours, radius
 * clamp(int(200/sqrt(max(1,dist))), 1, 20). rgba8/id_dist are the two outputs of o_render; out: W*H*4 bytes. */
void o_denoise(const uint8_t *rgba8, const int32_t *id_dist, int width, int height, uint8_t *out);

/* brute-force point query used to cross-check octreeFind: returns 1 and leaf
 * texels when `pos` is inside a leaf, 0 when empty; node AABB in mn/mx. */
int o_find_point(const o_scene *s, const int32_t pos[3], uint8_t leaf[8],
                 int32_t mn[3], int32_t mx[3]);

/* traversal-study hook: one record per in-world octreeFind (depth of the node found, depth the descent started at) */
typedef struct { uint32_t pixel; int16_t x, y, z; uint8_t found_depth, start_depth, leaf, pad; } o_find_rec;
void o_set_find_trace(o_find_rec *buf, size_t cap);
size_t o_find_trace_count(void);

uint64_t o_fnv1a64(const uint8_t *p, size_t n);

/* deterministic fp32 transcendental conventions shared (by restatement) with the kernels */
float o_det_expf(float x);
float o_det_sinf(float x);
float o_det_cosf(float x);
float o_det_powf(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
