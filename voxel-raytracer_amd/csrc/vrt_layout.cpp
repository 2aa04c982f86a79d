// vrt_layout.cpp -- see vrt_layout.h
#include "vrt_layout.h"

#include <deque>

namespace vrt {

namespace {
struct Pending { uint32_t texel; uint32_t depth; uint32_t record; };
constexpr uint32_t kMaxRecords = 1u << 23;  // a DFS-flattened tree has fewer records than texels, and texels < 2^23
constexpr uint32_t kMaxDepthInternal = 15;  // the shader's descent loop runs 16 iterations (raytracing.comp:161)
}  // namespace

bool build_layout(const uint8_t *texels, size_t used_bytes, Layout &out, std::string &err) {
    out = Layout();
    const size_t n_texels = texels ? used_bytes / 4 : 0;
    // zero-padded volume: anything past the used range reads as 0 (src/main.cpp:273-289)
    auto tex = [&](uint64_t i) -> uint32_t {
        if (i >= n_texels) return 0u;
        const uint8_t *p = texels + i * 4;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    };

    // a tree written by octree_texture() expands to fewer records than it has texels; streams that
    // share or cycle sub-trees are walked as far as 8x their size, then rejected
    const uint64_t limit = n_texels * 8 + 8 < kMaxRecords ? n_texels * 8 + 8 : kMaxRecords;
    out.records.push_back(Record{0u, 0u});
    std::deque<Pending> queue;
    queue.push_back(Pending{0u, 0u, 0u});
    out.level_start.push_back(0u);
    uint32_t level_of_next_alloc = 0;

    while (!queue.empty()) {
        Pending n = queue.front();
        queue.pop_front();
        const uint32_t header = tex(n.texel);
        const uint32_t base = header & 0x7fffffu;  // decodePointer drops bit 23 (raytracing.comp:89-96)
        uint32_t mask = header >> 24;
        // A node met on the 16th iteration can only report "empty child": its children are never read.
        if (n.depth >= kMaxDepthInternal) mask = 0;
        const uint32_t first_child = (uint32_t)out.records.size();
        uint32_t leaf_mask = 0, rank = 0;
        if (mask != 0 && n.depth + 1 > level_of_next_alloc) {
            level_of_next_alloc = n.depth + 1;
            out.level_start.push_back(first_child);
        }
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const uint32_t ptr = tex((uint64_t)base + rank);
            ++rank;
            const uint32_t addr = ptr & 0x7fffffu;
            if (out.records.size() >= limit) {
                err = "octree texel stream expands past the record limit (cyclic or corrupt pointers)";
                return false;
            }
            if (ptr & 0x800000u) {
                const uint32_t t0 = tex(addr), t1 = tex((uint64_t)addr + 1);
                out.records.push_back(Record{(t0 & 0x00ffffffu) | (t1 & 0xff000000u), t1 & 0x00ffffffu});
                leaf_mask |= 1u << ci;
                ++out.n_leaves;
                if (n.depth + 1 > out.max_depth) out.max_depth = n.depth + 1;
            } else {
                const uint32_t rec = (uint32_t)out.records.size();
                out.records.push_back(Record{0u, 0u});
                queue.push_back(Pending{addr, n.depth + 1, rec});
                if (n.depth + 1 > out.max_depth) out.max_depth = n.depth + 1;
            }
        }
        out.records[n.record] = Record{mask | (leaf_mask << 8), first_child};
        ++out.n_internal;
    }
    out.level_start.push_back((uint32_t)out.records.size());
    return true;
}

bool has_unit_internal_node(const std::vector<Record> &records, const int wmin[3], const int wmax[3]) {
    struct Item { uint32_t rec; int mn[3], mx[3]; };
    if (records.empty()) return false;
    std::vector<Item> stack;
    Item root;
    root.rec = 0;
    for (int k = 0; k < 3; ++k) { root.mn[k] = wmin[k]; root.mx[k] = wmax[k]; }
    stack.push_back(root);
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const uint32_t masks = records[it.rec].w0, base = records[it.rec].w1;
        if ((masks & 0xffu) == 0) continue;  // no children: never descended through
        if (it.mx[0] - it.mn[0] <= 1 && it.mx[1] - it.mn[1] <= 1 && it.mx[2] - it.mn[2] <= 1) return true;
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((masks >> ci) & 1u)) continue;
            const uint32_t idx = base + rank++;
            if ((masks >> (8 + ci)) & 1u) continue;  // leaf child
            if (idx >= records.size()) continue;
            Item ch;
            ch.rec = idx;
            for (int k = 0; k < 3; ++k) {
                const int mid = it.mn[k] + ((it.mx[k] - it.mn[k]) >> 1);
                const bool hi = (ci >> (2 - k)) & 1u;
                ch.mn[k] = hi ? mid : it.mn[k];
                ch.mx[k] = hi ? it.mx[k] : mid;
            }
            stack.push_back(ch);
        }
    }
    return false;
}

}  // namespace vrt
