// Address/UB-sanitizer harness for the host-side layout code (vrt_layout.cpp): lays out damaged texel streams, answers
// point lookups, plans / extracts / applies edit patches (some of them damaged too). CPU only.
//   python -c "import sys; sys.path.insert(0,'.'); import vrt_import; V=vrt_import.vrt(); w=V.World(); \
//              w.load_vox('tests/golden/maps/monu9.vox'); open('/tmp/monu9.tex','wb').write(bytes(w.flatten()[0]))"
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Ivoxel-raytracer_amd/csrc tools/asan_layout_harness.cpp \
//       voxel-raytracer_amd/csrc/vrt_layout.cpp -o /tmp/asan_layout && /tmp/asan_layout /tmp/monu9.tex
// Round 1: 3,000 streams (2,420 laid out, 580 refused), 58,392 patches applied, no sanitizer report.
// Round 2: + compact_records() after the patches of every third stream, lookups before == after.
#include "vrt_layout.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace vrt;
static std::vector<uint8_t> read_file(const char *p) {
    FILE *f = fopen(p, "rb"); std::vector<uint8_t> v; if (!f) return v;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (fread(v.data(), 1, n, f) != (size_t)n) v.clear(); fclose(f); return v;
}
int main(int argc, char **argv) {
    std::vector<uint8_t> base = read_file(argv[1]);
    std::mt19937 rng(7);
    const int wmin[3] = {-1023, -1023, -1023}, wmax[3] = {1024, 1024, 1024};
    int ok = 0, bad = 0, patched = 0, compacted = 0;
    long compared = 0;
    for (int it = 0; it < 3000; ++it) {
        std::vector<uint8_t> t = base;
        if (it % 5 == 0) { t.resize((rng() % 4000) / 4 * 4); for (auto &b : t) b = rng(); }
        else {
            int k = 1 + rng() % 6;
            for (int i = 0; i < k; ++i) t[rng() % t.size()] = rng();
            if (rng() % 4 == 0) t.resize((4 + rng() % (t.size() - 4)) / 4 * 4);
        }
        Layout lay; std::string err;
        if (!build_layout(t.data(), t.size(), lay, err)) { ++bad; continue; }
        ++ok;
        WideTree wt;
        const bool wide = !has_unit_internal_node(lay.records, wmin, wmax) && build_wide(lay.records, wmin, wmax, wt, err);
        for (int q = 0; q < 50; ++q) {
            int p[3] = {(int)(rng() % 2047) - 1023, (int)(rng() % 2047) - 1023, (int)(rng() % 2047) - 1023};
            if (q % 2) { p[0] = rng() % 128; p[1] = rng() % 128; p[2] = rng() % 128; }
            uint32_t w0, w1; int mn[3], mx[3];
            wide_find_host(lay.records, wide ? wt : WideTree(), wmin, wmax, p, w0, w1, mn, mx);
            PatchSite site;
            if (plan_patch(lay.records, wt, wide, wmin, wmax, p, 15, site)) {
                std::vector<Record> sub;
                if (extract_subtree(lay.records, site.path, site.depth, sub, (q & 2) ? p : nullptr, wmin, wmax)) {
                    if (!sub.empty() && q % 3 == 0) sub[rng() % sub.size()].w0 ^= 1u << (rng() % 16);   // damage the patch too
                    PatchRanges rg;
                    if (apply_patch(lay.records, wt, wide, site, sub.data(), sub.size(), rg, err)) ++patched;
                }
            }
        }
        // round 2: compaction of what the patches left behind (vrt_compact): the re-laid-out records must answer every
        // lookup as the patched ones did
        if (it % 3 == 0) {
            std::vector<Record> before = lay.records;
            compact_records(lay.records);
            ++compacted;
            WideTree none;
            for (int q = 0; q < 40; ++q) {
                int p[3] = {(int)(rng() % 2047) - 1023, (int)(rng() % 2047) - 1023, (int)(rng() % 2047) - 1023};
                if (q % 2) { p[0] = rng() % 128; p[1] = rng() % 128; p[2] = rng() % 128; }
                uint32_t a0, a1, b0, b1; int amn[3], amx[3], bmn[3], bmx[3];
                const int ka = wide_find_host(before, none, wmin, wmax, p, a0, a1, amn, amx);
                const int kb = wide_find_host(lay.records, none, wmin, wmax, p, b0, b1, bmn, bmx);
                ++compared;
                if (ka != kb || a0 != b0 || a1 != b1 || amn[0] != bmn[0] || amx[2] != bmx[2]) {
                    printf("compaction changed a lookup (stream %d, point %d %d %d)\n", it, p[0], p[1], p[2]);
                    return 1;
                }
            }
        }
    }
    printf("laid out %d refused %d patches applied %d compactions %d lookups compared %ld\n", ok, bad, patched, compacted, compared);
    return 0;
}
