// frame_multi.cpp -- the reference's frame loop on SEVERAL GPUs from one C++ process, no Python and no launcher:
// vrt_create_multi(n, device_ids) where src/main.cpp:432-474 creates its GL objects, vrt_multi_dispatch where
// src/main.cpp:946 calls glDispatchCompute. Every device traces its interleaved 8-row tiles of the frame; the frame is
// complete on the first device (peer stores over xGMI, or a pull gather).
//
//   g++ -std=c++17 -Iinclude examples/frame_multi.cpp -Lvoxel-raytracer_amd -lvrt_host -lvrt_hip -o frame_multi
//   ./frame_multi tests/golden/maps/dragon.vox 1920 1080 0,1,2,3,4,5,6,7 [gather]
#include <Camera.hpp>
#include <octree.hpp>
#include <voxReader.hpp>
#include <vrt.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static uint64_t fnv1a64(const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    const char *map = argc > 1 ? argv[1] : "tests/golden/maps/dragon.vox";
    const int W = argc > 2 ? atoi(argv[2]) : 1920, H = argc > 3 ? atoi(argv[3]) : 1080;
    std::vector<int> devices;
    for (const char *p = argc > 4 ? argv[4] : "0"; *p;) {
        devices.push_back((int)strtol(p, const_cast<char **>(&p), 10));
        if (*p == ',') ++p;
    }
    const int delivery = (argc > 5 && std::string(argv[5]) == "gather") ? VRT_DELIVER_GATHER : VRT_DELIVER_PEER_STORE;

    Octree *chunk0 = octree_create(NULL, {-1023, -1023, -1023}, {1024, 1024, 1024});   // src/main.cpp:478-480
    if (!load_vox_file(map, chunk0, 0, 0, 0)) { fprintf(stderr, "cannot load %s\n", map); return 1; }
    const size_t texels = _octree_texel_size(chunk0);
    size_t dim = (size_t)ceil(cbrt((double)texels)), bytes = 0;
    if (dim == 0) dim = 1;
    uint8_t *tex = octree_texture(chunk0, &bytes, dim);

    vrt_multi *m = nullptr;
    if (vrt_create_multi((int)devices.size(), devices.data(), &m) != VRT_OK) { fprintf(stderr, "%s\n", vrt_multi_last_error(nullptr)); return 1; }
    if (vrt_multi_upload_octree(m, tex, tex ? bytes : 0, (uint32_t)dim) != VRT_OK) { fprintf(stderr, "%s\n", vrt_multi_last_error(m)); return 1; }
    free(tex);

    Camera camera(vrtm::vec3(63.5f, 60.5f, 140.5f), vrtm::vec3(0.0f, 1.0f, 0.0f), -90.0f, -10.0f);
    float inv_proj[16], inv_view[16], cam[4];
    camera.FillDispatchBlock(W, H, inv_proj, inv_view, cam);   // src/main.cpp:808-813: inverse(perspective), inverse(lookAt), position
    vrt_multi_set_camera(m, inv_proj, inv_view, cam);

    void *d_rgba = nullptr, *d_id = nullptr;
    if (vrt_multi_frame_alloc(m, W, H, &d_rgba, &d_id) != VRT_OK) { fprintf(stderr, "%s\n", vrt_multi_last_error(m)); return 1; }
    const int frames = 200;
    for (int i = 0; i < 20; ++i) vrt_multi_dispatch(m, W, H, 8, VRT_MODE_PRIMARY, delivery, d_rgba, d_id);
    vrt_multi_synchronize(m);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i)
        if (vrt_multi_dispatch(m, W, H, 8, VRT_MODE_PRIMARY, delivery, d_rgba, d_id) != VRT_OK) { fprintf(stderr, "%s\n", vrt_multi_last_error(m)); return 1; }
    vrt_multi_synchronize(m);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<uint8_t> rgba((size_t)W * H * 4);
    std::vector<int32_t> idd((size_t)W * H * 2);
    vrt_device_read(vrt_multi_context(m, 0), d_rgba, rgba.data(), rgba.size(), nullptr);
    vrt_device_read(vrt_multi_context(m, 0), d_id, idd.data(), idd.size() * 4, nullptr);
    printf("{\"devices\": %zu, \"delivery\": \"%s\", \"width\": %d, \"height\": %d, \"ms_per_frame\": %.5f, \"Mrays_per_s\": %.1f, "
           "\"rgba_fnv1a64\": \"%016llx\", \"id_dist_fnv1a64\": \"%016llx\"}\n",
           devices.size(), delivery == VRT_DELIVER_GATHER ? "gather" : "peer_store", W, H, s / frames * 1e3, (double)W * H * frames / s / 1e6,
           (unsigned long long)fnv1a64(rgba.data(), rgba.size()), (unsigned long long)fnv1a64(idd.data(), idd.size() * 4));
    // and the frame the reference puts on screen: full path tracer + display pass, one band of rows per device (20-row
    // halos), only the displayed image crossing to device 0 -- into the rgba image's storage
    for (int i = 0; i < 5; ++i) vrt_multi_dispatch_frame(m, W, H, VRT_MODE_FULL, d_rgba);
    vrt_multi_synchronize(m);
    const auto t1 = std::chrono::steady_clock::now();
    const int shown_frames = 50;
    for (int i = 0; i < shown_frames; ++i)
        if (vrt_multi_dispatch_frame(m, W, H, VRT_MODE_FULL, d_rgba) != VRT_OK) { fprintf(stderr, "%s\n", vrt_multi_last_error(m)); return 1; }
    vrt_multi_synchronize(m);
    const double s2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    vrt_device_read(vrt_multi_context(m, 0), d_rgba, rgba.data(), rgba.size(), nullptr);
    printf("{\"devices\": %zu, \"displayed_frame\": \"full shader + display pass in row bands\", \"ms_per_frame\": %.5f, "
           "\"shown_fnv1a64\": \"%016llx\"}\n", devices.size(), s2 / shown_frames * 1e3, (unsigned long long)fnv1a64(rgba.data(), rgba.size()));
    vrt_multi_frame_free(m, d_rgba, d_id);
    vrt_destroy_multi(m);
    octree_delete(chunk0);
    return 0;
}
