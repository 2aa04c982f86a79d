// Address/UB-sanitizer harness for the host library's .vox loader, octree builder and flattener on damaged files. CPU only.
//   cd voxel-raytracer_amd/csrc/host && gcc -O1 -g -fsanitize=address,undefined -I../../../include -c vmm5.c color.c && \
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -I../../../include ../../../tools/asan_vox_harness.cpp octree.cpp \
//       voxReader.cpp vmm5.o color.o -o /tmp/asan_vox && /tmp/asan_vox ../../../tests/golden/maps/monu9.vox
// Round 1: 400 damaged files (184 loaded, 216 refused), no sanitizer report.
#include <octree.hpp>
#include <voxReader.hpp>
#include <cstdio>
#include <random>
#include <vector>
bool vrt_load_vox_memory(const uint8_t *data, size_t len, Octree *tree, int offsetX, int offsetY, int offsetZ, long *inserted);
static std::vector<uint8_t> read_file(const char *p) {
    FILE *f = fopen(p, "rb"); std::vector<uint8_t> v; if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
int main(int argc, char **argv) {
    std::vector<uint8_t> base = read_file(argv[1]);
    std::mt19937 rng(11);
    int ok = 0, bad = 0;
    for (int it = 0; it < 400; ++it) {
        std::vector<uint8_t> t = base;
        if (it % 7 == 0) { t.resize(rng() % 300); for (auto &b : t) b = rng(); }
        else {
            int k = 1 + rng() % 8;
            for (int i = 0; i < k; ++i) t[rng() % (it % 2 ? 200 : t.size())] = rng();   // headers and chunk sizes live up front
            if (rng() % 4 == 0) t.resize(1 + rng() % t.size());
        }
        Octree *root = octree_create(NULL, {-1023, -1023, -1023}, {1024, 1024, 1024});
        long inserted = 0;
        const bool r = vrt_load_vox_memory(t.data(), t.size(), root, 0, 0, 0, &inserted);
        r ? ++ok : ++bad;
        size_t used = 0;
        size_t texels = _octree_texel_size(root);
        size_t dim = 1; while (dim * dim * dim < texels) ++dim;
        uint8_t *tex = octree_texture(root, &used, dim);
        free(tex);
        octree_delete(root);
    }
    printf("loaded %d refused %d\n", ok, bad);
    return 0;
}
