// voxel.hpp -- what one voxel carries in the host API. Type and function names follow the reference's
// include/voxel.hpp so that application code written against it compiles here; the implementation lives in
// voxel-raytracer_amd/csrc/host/octree.cpp.
//
// How the values reach the GPU (flattening, octree.cpp of either code base): a leaf becomes two texels,
//   texel 0 = colour R, G, B (alpha of this texel is fixed at 255)
//   texel 1 = (uint8)(refraction * 85), (uint8)(illumination * 255), (uint8)(k * 255), colour alpha
// so `refraction` is an index of refraction in [0, 3] (3.0 = "opaque solid", the loader's default), `illumination`
// an emission strength in [0, 1] (non-zero voxels light themselves and cast no shadow), and `k` the Beer-Lambert
// absorption coefficient in [0, 1] used while a ray travels inside a translucent medium.
#ifndef VRT_VOXEL_HPP
#define VRT_VOXEL_HPP
#include <stdint.h>
extern "C" {
#include <color.h>
#include <vmm/ivec3.h>
}

typedef uint32_t Voxel_Type;  // index into the application's material table (voxels[] in voxReader.hpp)

// material of a voxel; see the byte encoding above
struct Voxel {
    float refraction, illumination, k;
};

// a voxel placed in the world: integer cell, packed colour (color.h: R in the top byte, alpha in the low one), material
struct Voxel_Object {
    IVector3 coord;
    ColorRGBA color;
    Voxel voxel;
};

// bundles the three parts (note the argument order: material, colour, cell)
Voxel_Object VoxelObjCreate(Voxel voxel, ColorRGBA color, IVector3 coord);

// Equality as the octree's merge step understands it: two materials are "the same" when refraction and illumination
// agree -- k is not looked at -- and two placed voxels when, in addition, their cells compare equal under the vector
// library's ivec3_equal_vec (which, as shipped with the reference, ignores y unless one of the two is 0; see
// vmm/ivec3.h). Colour takes no part in either.
bool voxel_compare(Voxel a, Voxel b);
bool voxel_obj_compare(Voxel_Object a, Voxel_Object b);

#endif
