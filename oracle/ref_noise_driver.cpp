// ref_noise_driver.cpp -- TEST INFRASTRUCTURE ONLY.
// Harness around the REFERENCE's own vendored include/FastNoiseLite.h (1.1.1), compiled in place from
// /root/reference/include; nothing is copied. Writes the height field of BASELINE config 4:
// the reference's commented-out terrain generator, src/main.cpp:487-503, scaled x4 to 1024 x 1024 columns --
//     FastNoiseLite noise(1337); noise.SetNoiseType(FastNoiseLite::NoiseType_Perlin);   // frequency 0.01 (default)
//     h(x, z) = (int)((noise.GetNoise((float)x, (float)z) + 1.0) * 33.0 * amp) + 120    // amp = 4
// as raw little-endian uint16, row z, column x, to the file named on the command line.
//     usage: ref_noise <out.u16> [size = 1024] [amp = 4] [seed = 1337]
#include <FastNoiseLite.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s out.u16 [size] [amp] [seed]\n", argv[0]); return 2; }
    const int size = argc > 2 ? std::atoi(argv[2]) : 1024;
    const double amp = argc > 3 ? std::atof(argv[3]) : 4.0;
    const int seed = argc > 4 ? std::atoi(argv[4]) : 1337;
    if (size < 1 || size > 4096) return 2;
    FastNoiseLite noise(seed);
    noise.SetNoiseType(FastNoiseLite::NoiseType_Perlin);
    std::vector<uint16_t> h((size_t)size * size);
    int lo = 1 << 30, hi = -(1 << 30);
    for (int z = 0; z < size; ++z)
        for (int x = 0; x < size; ++x) {
            // argument order of the reference's loop: GetNoise((float)j, (float)i) with voxel {j, h, i}
            const int v = (int)((noise.GetNoise((float)x, (float)z) + 1.0) * 33.0 * amp) + 120;
            if (v < 0 || v > 65535) { std::fprintf(stderr, "height %d out of range at (%d, %d)\n", v, x, z); return 1; }
            h[(size_t)z * size + x] = (uint16_t)v;
            if (v < lo) lo = v;
            if (v > hi) hi = v;
        }
    FILE *f = std::fopen(argv[1], "wb");
    if (!f || std::fwrite(h.data(), sizeof(uint16_t), h.size(), f) != h.size()) { std::perror(argv[1]); return 1; }
    std::fclose(f);
    std::printf("{\"size\": %d, \"amp\": %g, \"seed\": %d, \"min\": %d, \"max\": %d}\n", size, amp, seed, lo, hi);
    return 0;
}
