// voxel.cpp -- voxel value helpers (API of the reference's include/voxel.hpp / src/voxel.cpp).
#include <voxel.hpp>

Voxel_Object VoxelObjCreate(Voxel voxel, ColorRGBA color, IVector3 coord) {
    Voxel_Object o;
    o.coord = coord;
    o.color = color;
    o.voxel = voxel;
    return o;
}

// k does not take part (reference src/voxel.cpp:14-18)
bool voxel_compare(Voxel a, Voxel b) { return a.refraction == b.refraction && a.illumination == b.illumination; }

bool voxel_obj_compare(Voxel_Object a, Voxel_Object b) { return ivec3_equal_vec(a.coord, b.coord) && voxel_compare(a.voxel, b.voxel); }
