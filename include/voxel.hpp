// voxel.hpp -- voxel material/value types of the host API (reference: include/voxel.hpp).
#ifndef VRT_VOXEL_HPP
#define VRT_VOXEL_HPP
#include <stdint.h>
extern "C" {
#include <color.h>
#include <vmm/ivec3.h>
}

typedef uint32_t Voxel_Type;

struct Voxel {
    float refraction, illumination, k;
};

struct Voxel_Object {
    IVector3 coord;
    ColorRGBA color;
    Voxel voxel;
};

Voxel_Object VoxelObjCreate(Voxel voxel, ColorRGBA color, IVector3 coord);
bool voxel_compare(Voxel a, Voxel b);               // refraction + illumination only (k ignored)
bool voxel_obj_compare(Voxel_Object a, Voxel_Object b);

#endif
