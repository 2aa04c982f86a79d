// octree.hpp -- sparse voxel octree host API (reference: include/octree.hpp:15-30).
// Same node type and entry points as the reference so main.cpp-style callers
// compile unchanged; the implementation (voxel-raytracer_amd/csrc/host/octree.cpp)
// is this repository's own and produces a byte-identical octree_texture() stream.
#ifndef VRT_OCTREE_HPP
#define VRT_OCTREE_HPP
#include <stdint.h>
#include <stdlib.h>
#include <voxel.hpp>
extern "C" {
#include <color.h>
#include <vmm/ivec3.h>
#include <vmm/ray.h>
}

typedef struct _octree {
    Voxel_Object voxel;
    bool has_voxel;
    struct _octree **children, *parent;  // children: NULL or exactly 8 entries
    IVector3 left_bot_back, right_top_front;  // half-open AABB [min, max)
} Octree;

Octree *octree_new(void);
Octree *octree_create(Octree *parent, IVector3 left_bot_back, IVector3 right_top_front);
void octree_insert(Octree *tree, Voxel_Object voxel);
Voxel_Object octree_find(Octree *tree, IVector3 coord);
Octree *octree_ray_cast(Octree *root, Ray ray, Vector3 box_min, Vector3 box_max);
// calloc'd texel bytes (4 per texel), *arr_size = byte count; NULL/0 for an empty tree. Caller frees.
uint8_t *octree_texture(Octree *tree, size_t *arr_size, size_t tex_dim);
size_t _octree_texel_size(Octree *tree);
void octree_remove(Octree *tree, IVector3 coord);
void octree_delete(Octree *tree);

#endif
