// frame_main.cpp -- the reference's frame loop (src/main.cpp:478-485, 641, 807-813, 916-946, 951-967) reduced to
// its data path and written against THIS repository's headers exactly as main.cpp is written against the
// reference's: same type and function names (Octree, octree_create, load_vox_file, _octree_texel_size,
// octree_texture, Camera, octree_ray_cast, vec3_scalar_mul, make_color_rgba, VoxelObjCreate, octree_insert,
// octree_remove), with the GL calls replaced by the C-ABI of include/vrt.h.
//
//   g++ -std=c++17 -Iinclude examples/frame_main.cpp -Lvoxel-raytracer_amd -lvrt_host -lvrt_hip -o frame_main
//   ./frame_main tests/golden/maps/dragon.vox 320 180
#include <Camera.hpp>
#include <octree.hpp>
#include <voxReader.hpp>
#include <vrt.h>
#include <vrt_host.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#define WORLD_SIZE_X 1024
#define WORLD_SIZE_Y 1024
#define WORLD_SIZE_Z 1024

static vrt_ctx *g_vrt = nullptr;
static size_t tex_dim = 0;

static uint64_t fnv1a64(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// src/main.cpp:264-311
static void updateGPUTexture(Octree *tree) {
    size_t total_texels = _octree_texel_size(tree);
    tex_dim = (size_t)ceil(cbrt((double)total_texels));
    if (tex_dim == 0) tex_dim = 1;
    size_t arr_size_used = 0;
    uint8_t *texture_data = octree_texture(tree, &arr_size_used, tex_dim);
    if (vrt_upload_octree(g_vrt, texture_data, texture_data ? arr_size_used : 0, (uint32_t)tex_dim) != VRT_OK)
        std::cerr << "vrt_upload_octree: " << vrt_last_error(g_vrt) << std::endl;
    free(texture_data);
}

// after the host octree was edited at voxel v: replace the smallest enclosing sub-tree on the device instead of
// re-flattening and re-uploading everything; falls back to the reference's full upload
static void patchGPUVoxel(Octree *tree, IVector3 v) {
    vrt_patch plan;
    for (int max_depth = 15; max_depth >= 1; max_depth = plan.depth - 1) {
        if (vrt_patch_plan(g_vrt, v.x, v.y, v.z, max_depth, &plan) != VRT_OK) break;
        if (vrth_octree_node_state(tree, plan.path, plan.depth) != 2) continue;
        uint32_t *recs = nullptr;
        size_t n = 0;
        if (vrth_octree_path_records(tree, plan.path, plan.depth, v.x, v.y, v.z, &recs, &n) != 0) break;
        const int rc = vrt_patch_apply(g_vrt, &plan, recs, n);
        vrth_free(recs);
        if (rc == VRT_OK) {
            vrt_scene_info info;
            vrt_get_scene_info(g_vrt, &info);
            tex_dim = info.tex_dim;
            return;
        }
        break;
    }
    updateGPUTexture(tree);
}

int main(int argc, char **argv) {
    const char *map = argc > 1 ? argv[1] : "tests/golden/maps/dragon.vox";
    const int screenWidth = argc > 2 ? atoi(argv[2]) : 320, screenHeight = argc > 3 ? atoi(argv[3]) : 180;
    if (vrt_create(0, &g_vrt) != VRT_OK) {
        std::cerr << vrt_last_error(nullptr) << std::endl;
        return 2;
    }
    Octree *chunk0 = octree_create(NULL, {-WORLD_SIZE_X + 1, -WORLD_SIZE_Y + 1, -WORLD_SIZE_Z + 1},
                                   {WORLD_SIZE_X, WORLD_SIZE_Y, WORLD_SIZE_Z});
    if (!load_vox_file(map, chunk0, 0, 0, 0)) return 3;
    float voxelScale = 1.0f;
    updateGPUTexture(chunk0);

    Camera camera(vrtm::vec3(63.5f, 60.5f, 140.5f), vrtm::vec3(0.0f, 1.0f, 0.0f), -90.0f, -10.0f);
    vrt_params prm;
    vrt_default_params(&prm);
    std::vector<uint8_t> rgba((size_t)screenWidth * screenHeight * 4), shown(rgba.size());
    std::vector<int32_t> idDist((size_t)screenWidth * screenHeight * 2);

    for (int frame = 0; frame < 2; ++frame) {
        // CPU pick ray (src/main.cpp:824-833) -> highlighted voxel uniform
        Ray ray;
        ray.origin = vec3_scalar_mul({camera.Position.x, camera.Position.y, camera.Position.z}, voxelScale);
        ray.direction = {camera.Front.x, camera.Front.y, camera.Front.z};
        Octree *hitNode = octree_ray_cast(chunk0, ray, {0, 0, 0}, {(float)WORLD_SIZE_X, (float)WORLD_SIZE_Y, (float)WORLD_SIZE_Z});
        if (hitNode && hitNode->has_voxel) {
            prm.highlighted[0] = hitNode->voxel.coord.x; prm.highlighted[1] = hitNode->voxel.coord.y; prm.highlighted[2] = hitNode->voxel.coord.z;
        } else {
            prm.highlighted[0] = prm.highlighted[1] = prm.highlighted[2] = -1;
        }
        if (frame == 1 && hitNode && hitNode->has_voxel) {
            // a "destroy" click followed by a "build" click (src/main.cpp:843-914), each patched in place on the device
            const IVector3 gone = hitNode->voxel.coord;
            octree_remove(chunk0, gone);
            patchGPUVoxel(chunk0, gone);
            Voxel light = {3.0f, 1.0f, 0.0f};
            octree_insert(chunk0, VoxelObjCreate(light, make_color_rgba(255, 210, 210, 255), {60, 70, 40}));
            patchGPUVoxel(chunk0, {60, 70, 40});
        }
        float invProj[16], invView[16], camPos[4];
        camera.FillDispatchBlock(screenWidth, screenHeight, invProj, invView, camPos);
        vrt_set_camera(g_vrt, invProj, invView, camPos);
        vrt_set_params(g_vrt, &prm);
        // :941-946 dispatch + :951-967 display pass; frame 0 through the two separate calls, frame 1 through the fused one
        const int rc = frame == 0
            ? (vrt_dispatch(g_vrt, screenWidth, screenHeight, VRT_MODE_FULL, rgba.data(), idDist.data()) != VRT_OK
                   ? -1 : vrt_denoise_host(g_vrt, screenWidth, screenHeight, rgba.data(), idDist.data(), shown.data()))
            : vrt_dispatch_frame(g_vrt, screenWidth, screenHeight, VRT_MODE_FULL, shown.data(), rgba.data(), idDist.data());
        if (rc != VRT_OK) {
            std::cerr << vrt_last_error(g_vrt) << std::endl;
            return 4;
        }
        printf("frame %d tex_dim %zu highlighted %d %d %d rgba %016llx id %016llx shown %016llx\n", frame, tex_dim,
               prm.highlighted[0], prm.highlighted[1], prm.highlighted[2], (unsigned long long)fnv1a64(rgba.data(), rgba.size()),
               (unsigned long long)fnv1a64(idDist.data(), idDist.size() * 4), (unsigned long long)fnv1a64(shown.data(), shown.size()));
    }
    octree_delete(chunk0);
    vrt_destroy(g_vrt);
    return 0;
}
