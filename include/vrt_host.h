/*
 * vrt_host.h -- C-ABI over the host-side library (libvrt_host.so) for callers
 * that cannot include the C++ headers (ctypes / cgo / JNI style bindings).
 * It wraps, without adding behaviour, the kept host API of the reference:
 *   octree.hpp   (reference include/octree.hpp:22-30, src/octree.cpp)
 *   voxReader.hpp(reference include/voxReader.hpp:14, src/voxReader.cpp)
 *   Camera.hpp   (reference include/Camera.hpp:18-98; UBO math src/main.cpp:808-813)
 * plus a .vox writer and the deterministic synthetic scenes used by the
 * benchmark configurations whose inputs the reference does not ship.
 * All functions are plain C, return 0/positive on success and negative on error
 * unless stated otherwise.
 */
#ifndef VRT_HOST_H
#define VRT_HOST_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrth_world vrth_world; /* owns one octree root */

/* root AABB [min,max); NULL/NULL = the reference's chunk (-1023)^3 .. (1024)^3 (src/main.cpp:478-480) */
vrth_world *vrth_world_create(const int32_t *min3, const int32_t *max3);
void vrth_world_destroy(vrth_world *w);
void *vrth_world_root(vrth_world *w); /* Octree* for C++ callers */

/* load_vox_file / in-memory variant: returns 1 or 0 like the reference's bool */
int vrth_world_load_vox(vrth_world *w, const char *path, int ox, int oy, int oz);
int vrth_world_load_vox_mem(vrth_world *w, const uint8_t *data, size_t len, int ox, int oy, int oz, long *inserted);

/* octree_insert(VoxelObjCreate({refraction, illumination, k}, rgba, {x,y,z})) */
int vrth_world_insert(vrth_world *w, int x, int y, int z, uint32_t rgba, float refraction, float illumination, float k);
int vrth_world_insert_many(vrth_world *w, const int32_t *xyz, const uint32_t *rgba, size_t n, float refraction,
                           float illumination, float k);
int vrth_world_remove(vrth_world *w, int x, int y, int z);
/* octree_find: out7 = {coord.x, coord.y, coord.z, color, refraction bits, illumination bits, k bits} */
int vrth_world_find(vrth_world *w, int x, int y, int z, uint32_t out7[7]);
/* octree_ray_cast with box (0,0,0)-(1024,1024,1024) as the reference passes (src/main.cpp:827):
 * returns 1 and the hit node's voxel coord/has_voxel, 0 on miss */
int vrth_world_ray_cast(vrth_world *w, const float origin[3], const float dir[3], int32_t hit_coord[3], int *has_voxel);
/* the same for n rays from one origin (dirs: n x 3 floats), one octree_ray_cast each, in order: BASELINE config 1's frame loop
 * without a foreign-function call per pixel. hit[i] (optional) = 0 miss, 1 a node holding a voxel, 2 a node without one;
 * hit_coords (optional) n x 3. Returns the number of rays that returned a node, or -1. */
long vrth_world_ray_cast_many(vrth_world *w, const float origin[3], const float *dirs, size_t n, uint8_t *hit, int32_t *hit_coords);

size_t vrth_world_texel_count(vrth_world *w); /* _octree_texel_size */
/* updateGPUTexture's host half (src/main.cpp:264-271): texel stream + tex_dim.
 * *texels is malloc'd (release with vrth_free); empty world -> NULL, 0, dim 1 */
int vrth_world_flatten(vrth_world *w, uint8_t **texels, size_t *bytes, uint32_t *tex_dim);
/* EXTENSION: the device record array for vrt_upload_records() (vrt.h) straight from the pointer octree --
 * no texel stream, no 2^23-texel limit. *records is malloc'd (vrth_free), 2 uint32 per record.
 * Returns 0, or -2 when the root is itself a leaf (use the texel path then). */
int vrth_world_records(vrth_world *w, uint32_t **records, size_t *n_records, uint32_t *tex_dim);
/* the same for a caller that holds the Octree* itself (octree.hpp) */
int vrth_octree_records(void *octree_root, uint32_t **records, size_t *n_records, uint32_t *tex_dim);
/* EXTENSION, for vrt_patch_plan / vrt_patch_apply (vrt.h): the node reached from the root by `depth` child indices
 * (child i = (x >= mid) * 4 + (y >= mid) * 2 + (z >= mid), src/octree.cpp:46-76) as the flattening sees it --
 * 0 absent, 1 leaf, 2 internal -- and, for an internal node, its sub-tree as device records (that node = record 0,
 * child indices local to the sub-tree; vrth_free). */
int vrth_octree_node_state(void *octree_root, const uint8_t *path, int depth);
int vrth_octree_subtree_records(void *octree_root, const uint8_t *path, int depth, uint32_t **records, size_t *n_records);
/* the sub-tree for an edit of voxel (x, y, z): only the nodes that contain the voxel are walked and emitted, every
 * other internal child is a "keep" record {0xffffffff, 0xffffffff} that vrt_patch_apply resolves to what is there */
int vrth_octree_path_records(void *octree_root, const uint8_t *path, int depth, int x, int y, int z, uint32_t **records,
                             size_t *n_records);
int vrth_world_path_records(vrth_world *w, const uint8_t *path, int depth, int x, int y, int z, uint32_t **records,
                            size_t *n_records);
/* ... and for an edit of a whole box of voxels [lo, hi] (inclusive) under the node vrt_patch_plan_box names: the nodes that meet the box */
int vrth_octree_box_records(void *octree_root, const uint8_t *path, int depth, const int32_t lo[3], const int32_t hi[3], uint32_t **records,
                            size_t *n_records);
int vrth_world_box_records(vrth_world *w, const uint8_t *path, int depth, const int32_t lo[3], const int32_t hi[3], uint32_t **records,
                           size_t *n_records);
int vrth_world_node_state(vrth_world *w, const uint8_t *path, int depth);
int vrth_world_subtree_records(vrth_world *w, const uint8_t *path, int depth, uint32_t **records, size_t *n_records);
void vrth_free(void *p);

/* Camera(position, up=(0,1,0), yaw, pitch) -> the dispatch's Camera block for a width x height frame;
 * front3 (optional) receives Camera::Front */
int vrth_camera_block(const float pos[3], float yaw, float pitch, int width, int height, float inv_projection[16],
                      float inv_view[16], float camera_pos[4], float *front3);

/* MagicaVoxel writer (version 150: SIZE, XYZI, RGBA): xyzi = n * {x,y,z,colorIndex}, palette = 256 * {r,g,b,a} */
int vrth_write_vox(const char *path, int sx, int sy, int sz, const uint8_t *xyzi, size_t n, const uint8_t *palette_rgba);
/* same bytes into a malloc'd buffer (vrth_free) */
int vrth_encode_vox(int sx, int sy, int sz, const uint8_t *xyzi, size_t n, const uint8_t *palette_rgba, uint8_t **out,
                    size_t *out_len);

/* deterministic synthetic scenes:
 *  config 1 "custom.vox" stand-in: 64^3 model, floor slab + sphere -> malloc'd .vox bytes */
int vrth_make_custom_vox(uint8_t **out, size_t *out_len);
/*  config 4: the reference's commented-out terrain generator (src/main.cpp:487-503) over a height field given as
 *  data (size_x * size_z uint16, row z, column x): columns x in [x0, x0+nx), z in [z0, z0+nz) are filled for
 *  y in [max(floor_y, h - band), h) -- two lowest voxels STONE, top one DIRT, the rest GRASS (main.cpp:220-259) */
int vrth_world_fill_heights(vrth_world *w, const uint16_t *heights, int size_x, int size_z, int x0, int z0, int nx, int nz,
                            int band, int floor_y);

uint64_t vrth_fnv1a64(const uint8_t *p, size_t n);
const char *vrth_version(void);

#ifdef __cplusplus
}
#endif
#endif
