// voxReader.hpp -- MagicaVoxel .vox loader of the host API (reference: include/voxReader.hpp:14).
#ifndef VRT_VOXREADER_HPP
#define VRT_VOXREADER_HPP
extern "C" {
#include <color.h>
}
#include <octree.hpp>
#include <voxel.hpp>

// Material / colour tables the reference's application defines (src/main.cpp:220-259)
// and its loader reads (voxels[0] is the material given to every loaded voxel).
extern Voxel voxels[];
extern ColorRGBA voxelColors[];

bool load_vox_file(const char *filename, Octree *tree, int offsetX, int offsetY, int offsetZ);

#endif
