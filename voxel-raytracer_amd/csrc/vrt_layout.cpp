// vrt_layout.cpp -- see vrt_layout.h
#include "vrt_layout.h"

#include <algorithm>
#include <cstring>

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

namespace {

struct Box { int mn[3], mx[3]; };

// aligned cube of even log2 side >= 2 (the shape a wide node covers)
bool aligned_even_cube(const Box &b, int &shift) {
    const int sx = b.mx[0] - b.mn[0];
    if (sx != b.mx[1] - b.mn[1] || sx != b.mx[2] - b.mn[2]) return false;
    if (sx < 4 || sx > (1 << 30) || (sx & (sx - 1)) != 0) return false;
    if (((b.mn[0] | b.mn[1] | b.mn[2]) & (sx - 1)) != 0) return false;
    int s = 0;
    while ((1 << s) < sx) ++s;
    if (s & 1) return false;
    shift = s;
    return true;
}

Box child_box(const Box &b, uint32_t ci) {  // the shader's split: mid = min + (max - min) / 2
    Box c;
    for (int k = 0; k < 3; ++k) {
        const int mid = b.mn[k] + ((b.mx[k] - b.mn[k]) >> 1);
        const bool hi = (ci >> (2 - k)) & 1u;
        c.mn[k] = hi ? mid : b.mn[k];
        c.mx[k] = hi ? b.mx[k] : mid;
    }
    return c;
}

enum { kAbsent = 0, kLeaf = 1, kInternal = 2 };
int child_of(const std::vector<Record> &recs, uint32_t rec, uint32_t ci, uint32_t &idx) {
    const uint32_t masks = recs[rec].w0, bit = 1u << ci;
    if (!(masks & bit)) return kAbsent;
    idx = recs[rec].w1 + (uint32_t)__builtin_popcount(masks & 0xffu & (bit - 1u));
    if (idx >= recs.size()) return kAbsent;  // cannot happen for layouts build_layout() produced
    return (masks & (bit << 8)) ? kLeaf : kInternal;
}

// The property word a wide cell carries: refraction byte forced to 0 when the leaf's alpha byte is 0, so
// that the kernels' medium test (alpha > 0 && refraction > 0, raytracing.comp:318-319) is one AND. The
// shader overwrites the properties of alpha-0 voxels before reading them (comp:503-504), so the byte is
// unobservable there; the lookup at the eye, which is not, reads the record array instead.
inline uint32_t leaf_props(const Record &leaf) {
    const uint32_t w1 = leaf.w1 & 0x00ffffffu;
    return (leaf.w0 >> 24) != 0u ? w1 : (w1 & 0x00ffff00u);
}

// Reuse (patches): `old_node` >= 0 is the wide node that stood for this octree node before the edit. A grandchild
// record with an index below `frozen_below` (and not `rewritten`, the one record a patch changes in place) is an
// old, immutable record, so its whole sub-tree is unchanged: if the old wide node had a child built from that very
// record, that child is kept instead of being rebuilt.
bool build_wide_node(const std::vector<Record> &recs, uint32_t rec, int shift, WideTree &out, uint32_t &node, std::string &why,
                     long old_node = -1, size_t frozen_below = 0, uint32_t rewritten = 0xffffffffu) {
    node = out.n_nodes++;
    out.cells.resize((size_t)out.n_nodes * 64, WideCell{0u, 0u});
    out.node_record.resize(out.n_nodes, 0u);
    out.node_record[node] = rec;
    for (uint32_t cell = 0; cell < 64; ++cell) {
        const uint32_t cx = (cell >> 4) & 3u, cy = (cell >> 2) & 3u, cz = cell & 3u;
        const uint32_t hi = ((cx >> 1) << 2) | ((cy >> 1) << 1) | (cz >> 1);
        const uint32_t lo = ((cx & 1u) << 2) | ((cy & 1u) << 1) | (cz & 1u);
        uint32_t i1 = 0, i2 = 0;
        WideCell c{0u, 0u};
        const int k1 = child_of(recs, rec, hi, i1);
        if (k1 == kAbsent) {
            c.w1 = (uint32_t)(shift - 1) << 24;
        } else if (k1 == kLeaf) {
            c.w0 = recs[i1].w0;
            c.w1 = leaf_props(recs[i1]) | ((uint32_t)(shift - 1) << 24);
        } else {
            const int k2 = child_of(recs, i1, lo, i2);
            if (k2 == kAbsent) {
                c.w1 = (uint32_t)(shift - 2) << 24;
            } else if (k2 == kLeaf) {
                c.w0 = recs[i2].w0;
                c.w1 = leaf_props(recs[i2]) | ((uint32_t)(shift - 2) << 24);
            } else {
                if (shift - 2 < 2) {
                    why = "an internal node of unit size lies inside an aligned cube";
                    return false;
                }
                uint32_t child = 0;
                long old_child = -1;
                if (old_node >= 0) {
                    const WideCell oc = out.cells[(size_t)old_node * 64 + cell];
                    if (oc.w1 == kWideInternal && oc.w0 < out.node_record.size()) old_child = (long)oc.w0;
                }
                if (old_child >= 0 && i2 < frozen_below && i2 != rewritten && out.node_record[(size_t)old_child] == i2) {
                    child = (uint32_t)old_child;  // unchanged sub-tree: keep its wide nodes
                } else if (!build_wide_node(recs, i2, shift - 2, out, child, why, old_child, frozen_below, rewritten)) {
                    return false;
                }
                c.w0 = child;
                c.w1 = kWideInternal;
            }
        }
        out.cells[(size_t)node * 64 + cell] = c;
    }
    return true;
}

bool collect_roots(const std::vector<Record> &recs, uint32_t rec, const Box &box, int depth, WideTree &out, std::string &why) {
    if ((recs[rec].w0 & 0xffu) == 0 || depth > 16) return true;
    int shift = 0;
    if (aligned_even_cube(box, shift)) {
        if ((int)out.roots.size() >= kMaxWideRoots) {
            why = "more aligned sub-trees than the kernel's root table holds";
            return false;
        }
        uint32_t node = 0;
        if (!build_wide_node(recs, rec, shift, out, node, why)) return false;
        out.roots.push_back(WideRoot{rec, node, shift, {box.mn[0], box.mn[1], box.mn[2]}});
        return true;
    }
    for (uint32_t ci = 0; ci < 8; ++ci) {
        uint32_t idx = 0;
        if (child_of(recs, rec, ci, idx) == kInternal)
            if (!collect_roots(recs, idx, child_box(box, ci), depth + 1, out, why)) return false;
    }
    return true;
}

}  // namespace

bool build_wide(const std::vector<Record> &records, const int wmin[3], const int wmax[3], WideTree &out, std::string &why) {
    out = WideTree();
    if (records.empty()) return true;
    if (has_unit_internal_node(records, wmin, wmax)) {
        why = "an internal node of unit size";
        return false;
    }
    Box world;
    for (int k = 0; k < 3; ++k) { world.mn[k] = wmin[k]; world.mx[k] = wmax[k]; }
    return collect_roots(records, 0, world, 0, out, why);
}

bool content_only_in_root0(const std::vector<Record> &records, const WideTree &wt) {
    if (wt.roots.size() != 1) return false;
    uint32_t cur = 0;
    const uint32_t target = wt.roots[0].record;
    for (int depth = 0; cur != target && depth < 16; ++depth) {
        if (cur >= records.size()) return false;
        const Record &r = records[cur];
        const uint32_t mask = r.w0 & 0xffu, leaves = (r.w0 >> 8) & 0xffu;
        if (mask == 0u || (mask & (mask - 1u)) != 0u || (leaves & mask) != 0u) return false;
        cur = r.w1;
    }
    return cur == target;
}

void tighten_root0(const WideTree &wt, const int (*eyes)[3], int n, int min_shift, uint32_t &node, int &shift, int origin[3]) {
    for (;;) {
        const int cs = shift - 2;
        if (cs < min_shift || (size_t)node * 64 + 64 > wt.cells.size()) return;
        int only = -1, count = 0;
        for (int cell = 0; cell < 64 && count < 2; ++cell) {
            const WideCell &wc = wt.cells[(size_t)node * 64 + (size_t)cell];
            if ((wc.w1 & kWideInternal) != 0u || (wc.w0 | (wc.w1 & 0x00ffffffu)) != 0u) { ++count; only = cell; }
        }
        if (count != 1) return;
        const WideCell &wc = wt.cells[(size_t)node * 64 + (size_t)only];
        if ((wc.w1 & kWideInternal) == 0u) return;
        const int o[3] = {origin[0] + (((only >> 4) & 3) << cs), origin[1] + (((only >> 2) & 3) << cs), origin[2] + ((only & 3) << cs)};
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k)
                if (eyes[i][k] < o[k] || eyes[i][k] >= o[k] + (1 << cs)) return;
        node = wc.w0; shift = cs;
        for (int k = 0; k < 3; ++k) origin[k] = o[k];
    }
}

int wide_find_host(const std::vector<Record> &records, const WideTree &wt, const int wmin[3], const int wmax[3],
                   const int p[3], uint32_t &w0, uint32_t &w1, int mn[3], int mx[3]) {
    Box b;
    for (int k = 0; k < 3; ++k) { b.mn[k] = wmin[k]; b.mx[k] = wmax[k]; }
    w0 = w1 = 0;
    uint32_t rec = 0;
    for (int i = 0; i < 16; ++i) {
        int shift = 0;
        const WideRoot *root = nullptr;
        if ((records[rec].w0 & 0xffu) != 0 && aligned_even_cube(b, shift))
            for (const WideRoot &r : wt.roots)
                if (r.record == rec) root = &r;
        if (root) {
            uint32_t node = root->node;
            int s = root->shift;
            for (;;) {
                const int cs = s - 2;
                const uint32_t ci = ((((uint32_t)p[0] >> cs) & 3u) << 4) | ((((uint32_t)p[1] >> cs) & 3u) << 2) | (((uint32_t)p[2] >> cs) & 3u);
                const WideCell c = wt.cells[(size_t)node * 64 + ci];
                if (c.w1 & kWideInternal) { node = c.w0; s = cs; continue; }
                const int t = (int)((c.w1 >> 24) & 31u);
                for (int k = 0; k < 3; ++k) { mn[k] = p[k] & -(1 << t); mx[k] = mn[k] + (1 << t); }
                w0 = c.w0; w1 = c.w1 & 0x00ffffffu;
                // a leaf whose alpha and words are all zero is indistinguishable from empty space for every consumer
                return (c.w0 | w1) != 0 ? 1 : 0;
            }
        }
        uint32_t ci = 0;
        for (int k = 0; k < 3; ++k) {
            const int mid = b.mn[k] + ((b.mx[k] - b.mn[k]) >> 1);
            if (p[k] >= mid) ci |= 1u << (2 - k);
        }
        b = child_box(b, ci);
        for (int k = 0; k < 3; ++k) { mn[k] = b.mn[k]; mx[k] = b.mx[k]; }
        uint32_t idx = 0;
        const int kind = child_of(records, rec, ci, idx);
        if (kind == kAbsent) return 0;
        if (kind == kLeaf) { w0 = records[idx].w0; w1 = leaf_props(records[idx]); return 1; }
        rec = idx;
    }
    return 0;
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

void eye_lookup(const std::vector<Record> &recs, const int wmin[3], const int wmax[3], const int p[3], uint32_t &w0,
                uint32_t &w1) {
    w0 = w1 = 0u;
    if (recs.empty()) return;
    Box b;
    for (int k = 0; k < 3; ++k) {
        if (p[k] < wmin[k] || p[k] >= wmax[k]) return;
        b.mn[k] = wmin[k];
        b.mx[k] = wmax[k];
    }
    uint32_t rec = 0;
    for (int i = 0; i < 16; ++i) {
        uint32_t ci = 0;
        for (int k = 0; k < 3; ++k) {
            const int mid = b.mn[k] + ((b.mx[k] - b.mn[k]) >> 1);
            if (p[k] >= mid) ci |= 1u << (2 - k);
        }
        b = child_box(b, ci);
        uint32_t idx = 0;
        const int kind = child_of(recs, rec, ci, idx);
        if (kind == kAbsent) return;
        if (kind == kLeaf) { w0 = recs[idx].w0; w1 = recs[idx].w1; return; }
        rec = idx;
    }
}

bool first_find(const WideTree &wide, const int wmin[3], const int wmax[3], const int p[3], int anchor_shift, FirstFind &out) {
    if (wide.roots.empty()) return false;
    const WideRoot &r = wide.roots[0];
    for (int k = 0; k < 3; ++k) {
        if (p[k] < wmin[k] || p[k] >= wmax[k]) return false;
        if ((uint32_t)(p[k] ^ r.origin[k]) >> (r.shift & 31)) return false;
    }
    uint32_t node = r.node;
    int s = r.shift;
    out.anode = node;
    out.as = s;
    for (;;) {
        const int cs = s - 2;
        const uint32_t ci = ((((uint32_t)p[0] >> cs) & 3u) << 4) | ((((uint32_t)p[1] >> cs) & 3u) << 2) | (((uint32_t)p[2] >> cs) & 3u);
        const WideCell c = wide.cells[(size_t)node * 64 + ci];
        if (!(c.w1 & kWideInternal)) {
            out.w0 = c.w0;
            out.w1 = c.w1;
            out.node = node;
            out.s = s;
            return true;
        }
        node = c.w0;
        s = cs;
        if (cs == anchor_shift) { out.anode = node; out.as = cs; }
        if (cs < 2 || (size_t)node * 64 + 63 >= wide.cells.size()) return false;
    }
}

size_t stream_texels(const Record *recs, size_t n, uint32_t top) {
    if (top >= n) return 0;
    size_t total = 0;
    std::vector<uint32_t> todo(1, top);  // internal records still to count
    while (!todo.empty()) {
        const uint32_t r = todo.back();
        todo.pop_back();
        const uint32_t mask = recs[r].w0 & 0xffu, leaf_mask = (recs[r].w0 >> 8) & 0xffu;
        const uint32_t n_child = (uint32_t)__builtin_popcount(mask);
        total += 1 + n_child;
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t idx = (size_t)recs[r].w1 + rank++;
            if (idx >= n) continue;
            if ((leaf_mask >> ci) & 1u) total += 2;
            else todo.push_back((uint32_t)idx);
        }
    }
    return total;
}

bool extract_subtree(const std::vector<Record> &recs, const uint8_t *path, int depth, std::vector<Record> &sub,
                     const int *toward, const int *wmin, const int *wmax) {
    sub.clear();
    if (recs.empty()) return false;
    Box box{{0, 0, 0}, {0, 0, 0}};
    if (toward)
        for (int k = 0; k < 3; ++k) { box.mn[k] = wmin[k]; box.mx[k] = wmax[k]; }
    uint32_t rec = 0;
    for (int d = 0; d < depth; ++d) {
        uint32_t idx = 0;
        if (path[d] > 7 || child_of(recs, rec, path[d], idx) != kInternal) return false;
        rec = idx;
        if (toward) box = child_box(box, path[d]);
    }
    struct Item { uint32_t src; size_t at; Box box; };
    std::vector<Item> queue(1, Item{rec, 0, box});
    sub.push_back(recs[rec]);
    for (size_t head = 0; head < queue.size(); ++head) {
        const Item it = queue[head];
        const Record r = recs[it.src];
        const uint32_t mask = r.w0 & 0xffu, leaf_mask = (r.w0 >> 8) & 0xffu;
        sub[it.at].w1 = (uint32_t)sub.size();
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t idx = (size_t)r.w1 + rank++;
            if (idx >= recs.size()) return false;
            if ((leaf_mask >> ci) & 1u) { sub.push_back(recs[idx]); continue; }
            Box cb = it.box;
            bool expand = true;
            if (toward) {
                cb = child_box(it.box, ci);
                for (int k = 0; k < 3; ++k) expand = expand && toward[k] >= cb.mn[k] && toward[k] < cb.mx[k];
            }
            if (expand) {
                queue.push_back(Item{(uint32_t)idx, sub.size(), cb});
                sub.push_back(recs[idx]);
            } else {
                sub.push_back(Record{kKeep, kKeep});
            }
        }
    }
    return true;
}

bool plan_patch(const std::vector<Record> &recs, const WideTree &wide, bool wide_in_use, const int wmin[3],
                const int wmax[3], const int p[3], int max_depth, PatchSite &site) {  // wide_in_use: by value, adjusted below
    if (recs.empty()) return false;
    if (wide.roots.empty()) wide_in_use = false;  // nothing aligned in this world: the kernels walk the records alone
    for (int k = 0; k < 3; ++k)
        if (p[k] < wmin[k] || p[k] >= wmax[k]) return false;
    Box box;
    for (int k = 0; k < 3; ++k) { box.mn[k] = wmin[k]; box.mx[k] = wmax[k]; }
    uint32_t rec = 0;
    int depth = 0;
    long wnode = -1;        // wide node whose octree node is being descended (two octree levels per wide node)
    int wshift = 0, level = 0, root_index = -1;
    uint32_t hi = 0, parent_node = 0, parent_cell = 0;
    bool found = false;
    uint8_t path[16] = {};
    for (;;) {
        if (wide_in_use && wnode < 0)
            for (size_t i = 0; i < wide.roots.size(); ++i)
                if (wide.roots[i].record == rec) {
                    wnode = (long)wide.roots[i].node;
                    wshift = wide.roots[i].shift;
                    level = 0;
                    root_index = (int)i;
                }
        if (depth >= 1 && depth <= max_depth && depth <= 15) {
            const bool at_wide_node = wnode >= 0 && level == 0;
            if (!wide_in_use || at_wide_node) {
                found = true;
                site.depth = depth;
                std::memcpy(site.path, path, sizeof path);
                site.record = rec;
                site.shift = at_wide_node ? wshift : -1;
                site.root_index = at_wide_node ? root_index : -1;
                site.parent_node = parent_node;
                site.parent_cell = parent_cell;
            }
        }
        if (depth >= 15 || depth >= max_depth) break;
        uint32_t ci = 0;
        for (int k = 0; k < 3; ++k) {
            const int mid = box.mn[k] + ((box.mx[k] - box.mn[k]) >> 1);
            if (p[k] >= mid) ci |= 1u << (2 - k);
        }
        uint32_t idx = 0;
        if (child_of(recs, rec, ci, idx) != kInternal) break;
        if (wnode >= 0) {
            if (level == 0) {
                hi = ci;
                level = 1;
            } else {
                const uint32_t cx = (((hi >> 2) & 1u) << 1) | ((ci >> 2) & 1u), cy = (((hi >> 1) & 1u) << 1) | ((ci >> 1) & 1u),
                               cz = ((hi & 1u) << 1) | (ci & 1u);
                const uint32_t cell = (cx << 4) | (cy << 2) | cz;
                const WideCell c = wide.cells[(size_t)wnode * 64 + cell];
                if (c.w1 != kWideInternal) break;  // layouts out of step: leave the deeper levels alone
                parent_node = (uint32_t)wnode;
                parent_cell = cell;
                root_index = -1;
                wnode = (long)c.w0;
                wshift -= 2;
                level = 0;
            }
        }
        path[depth] = (uint8_t)ci;
        rec = idx;
        box = child_box(box, ci);
        ++depth;
    }
    return found;
}

bool apply_patch(std::vector<Record> &recs, WideTree &wide, bool wide_in_use, const PatchSite &site, const Record *sub,
                 size_t n_sub, PatchRanges &out, std::string &why) {
    out = PatchRanges();
    out.records_appended_from = recs.size();
    out.cells_appended_from = wide.cells.size();
    if (!sub || n_sub < 1 || n_sub > (1u << 30)) { why = "empty sub-tree"; return false; }
    if (site.record >= recs.size()) { why = "patch site out of range"; return false; }
    // structural check (as vrt_upload_records): every child block lies after its parent and inside the sub-tree
    std::vector<uint8_t> kind(n_sub, 0);  // 1 internal, 2 leaf, 3 internal and unchanged (kKeep)
    kind[0] = 1;
    for (size_t i = 0; i < n_sub; ++i) {
        if (kind[i] != 1) continue;
        const uint32_t mask = sub[i].w0 & 0xffu, leaf_mask = (sub[i].w0 >> 8) & 0xffu, base = sub[i].w1;
        const uint32_t n_child = (uint32_t)__builtin_popcount(mask);
        if (n_child == 0) continue;
        if (base <= i || (size_t)base + n_child > n_sub) { why = "sub-tree child index out of order or range"; return false; }
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t idx = (size_t)base + rank++;
            if (kind[idx] != 0) { why = "a sub-tree record has two parents"; return false; }
            kind[idx] = ((leaf_mask >> ci) & 1u) ? 2 : ((sub[idx].w0 == kKeep && sub[idx].w1 == kKeep) ? 3 : 1);  // 3: kept as it was
        }
    }
    // old record standing where each new record stands (same position under A, same kind), if any
    constexpr uint32_t kNone = 0xffffffffu;
    std::vector<uint32_t> old_of(n_sub, kNone);
    old_of[0] = site.record;
    for (size_t j = 0; j < n_sub; ++j) {
        if (kind[j] != 1 || old_of[j] == kNone) continue;
        const uint32_t mask = sub[j].w0 & 0xffu, leaf_mask = (sub[j].w0 >> 8) & 0xffu;
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t sj = (size_t)sub[j].w1 + rank++;
            uint32_t oi = 0;
            const int ok = child_of(recs, old_of[j], ci, oi);
            if (ok == (((leaf_mask >> ci) & 1u) ? kLeaf : kInternal)) old_of[sj] = oi;
            else if (kind[sj] == 3) { why = "a kept child has no internal counterpart in the uploaded tree"; return false; }
        }
    }
    // the stream's texel count: what the expanded part of the new sub-tree takes minus what it replaces (kept
    // sub-trees count on neither side; the pointer texel to one belongs to its parent)
    {
        std::vector<uint32_t> kept;
        size_t t_new = 0;
        for (size_t j = 0; j < n_sub; ++j) {
            if (kind[j] == 3) {
                if (old_of[j] == kNone) { why = "a kept child has no counterpart in the uploaded tree"; return false; }
                kept.push_back(old_of[j]);
            }
            if (kind[j] != 1) continue;
            const uint32_t mask = sub[j].w0 & 0xffu, leaf_mask = (sub[j].w0 >> 8) & mask;
            t_new += 1 + (size_t)__builtin_popcount(mask) + 2 * (size_t)__builtin_popcount(leaf_mask);
        }
        std::sort(kept.begin(), kept.end());
        size_t t_old = 0;
        std::vector<uint32_t> todo(1, site.record);
        while (!todo.empty()) {
            const uint32_t r = todo.back();
            todo.pop_back();
            const uint32_t mask = recs[r].w0 & 0xffu, leaf_mask = (recs[r].w0 >> 8) & 0xffu;
            t_old += 1 + (size_t)__builtin_popcount(mask);
            uint32_t rank = 0;
            for (uint32_t ci = 0; ci < 8; ++ci) {
                if (!((mask >> ci) & 1u)) continue;
                const size_t idx = (size_t)recs[r].w1 + rank++;
                if (idx >= recs.size()) continue;
                if ((leaf_mask >> ci) & 1u) t_old += 2;
                else if (!std::binary_search(kept.begin(), kept.end(), (uint32_t)idx)) todo.push_back((uint32_t)idx);
            }
        }
        out.texel_delta = (long)t_new - (long)t_old;
    }
    // same[j]: the new sub-tree under j equals the old one under old_of[j] (bottom-up: children have larger indices)
    std::vector<uint8_t> same(n_sub, 0);
    for (size_t j = n_sub; j-- > 0;) {
        if (old_of[j] == kNone) continue;
        const Record &o = recs[old_of[j]];
        if (kind[j] == 3) { same[j] = 1; continue; }
        if (kind[j] == 2) { same[j] = (o.w0 == sub[j].w0 && o.w1 == sub[j].w1); continue; }
        if (o.w0 != sub[j].w0) continue;
        bool all = true;
        const uint32_t n_child = (uint32_t)__builtin_popcount(sub[j].w0 & 0xffu);
        for (uint32_t r = 0; r < n_child && all; ++r) all = same[(size_t)sub[j].w1 + r] != 0;
        same[j] = all;
    }
    // emit top-down: a changed internal node gets a new child block; unchanged children keep their old records'
    // contents (and with them their old sub-trees), changed internal children are filled in when their turn comes
    std::vector<size_t> where(n_sub, 0);  // global index of the record of sub[j]
    where[0] = site.record;
    for (size_t j = 0; j < n_sub; ++j) {
        if (kind[j] != 1 || same[j] || (j != 0 && where[j] == 0)) continue;
        const uint32_t n_child = (uint32_t)__builtin_popcount(sub[j].w0 & 0xffu);
        const size_t block = recs.size();
        for (uint32_t r = 0; r < n_child; ++r) {
            const size_t sj = (size_t)sub[j].w1 + r;
            if (kind[sj] == 2) recs.push_back(sub[sj]);
            else if (same[sj]) { const Record keep = recs[old_of[sj]]; recs.push_back(keep); }
            else { recs.push_back(Record{sub[sj].w0, 0u}); where[sj] = block + r; }
        }
        recs[where[j]].w0 = sub[j].w0;
        recs[where[j]].w1 = n_child ? (uint32_t)block : 0u;
    }
    if (!wide_in_use || wide.roots.empty()) return true;
    if (site.shift < 2) { out.wide_invalid = true; return true; }
    if (same[0]) return true;  // nothing changed below A
    uint32_t node = 0;
    std::string wide_why;
    const long old_node = site.root_index >= 0 ? (long)wide.roots[(size_t)site.root_index].node
                                               : (long)wide.cells[(size_t)site.parent_node * 64 + site.parent_cell].w0;
    if (!build_wide_node(recs, site.record, site.shift, wide, node, wide_why, old_node, out.records_appended_from, site.record)) {
        out.wide_invalid = true;  // e.g. an internal node of unit size appeared: let the dispatcher re-derive everything
        return true;
    }
    if (site.root_index >= 0) {
        wide.roots[(size_t)site.root_index].node = node;
    } else {
        WideCell &c = wide.cells[(size_t)site.parent_node * 64 + site.parent_cell];
        c.w0 = node;
        c.w1 = kWideInternal;
        out.cell_repointed = true;
    }
    return true;
}

void compact_records(std::vector<Record> &records) {
    if (records.empty()) return;
    std::vector<Record> out;
    out.reserve(records.size());
    std::vector<uint32_t> old_of;      // old index of out[i]
    std::vector<uint8_t> is_leaf;
    out.push_back(records[0]);
    old_of.push_back(0u);
    is_leaf.push_back(0);
    for (size_t i = 0; i < out.size(); ++i) {   // breadth first: out grows while it is walked
        if (is_leaf[i]) continue;
        const Record r = records[old_of[i]];
        const uint32_t mask = r.w0 & 0xffu, leaf_mask = (r.w0 >> 8) & 0xffu;
        const uint32_t n_child = (uint32_t)__builtin_popcount(mask);
        if (n_child == 0 || (size_t)r.w1 + n_child > records.size()) { out[i].w0 = r.w0 & ~0xffffu; out[i].w1 = 0; continue; }
        out[i].w1 = (uint32_t)out.size();
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const uint32_t old = r.w1 + rank++;
            out.push_back(records[old]);
            old_of.push_back(old);
            is_leaf.push_back((uint8_t)((leaf_mask >> ci) & 1u));
        }
    }
    records.swap(out);
}

// True when pathTrace cannot take its translucent branch (comp:546-572) or absorb (comp:482-486, 512-516) anywhere in this tree for
// an eye in empty space: every leaf has alpha 0 (never a hit: its medium byte reads as empty space) or alpha 255 with a refraction
// byte that makes it a surface (not 0 and not 85, which the hit test cannot tell from empty space).
bool tree_is_opaque(const std::vector<Record> &rec) {
    if (rec.empty()) return false;
    std::vector<uint32_t> todo{0u};
    while (!todo.empty()) {
        const uint32_t i = todo.back();
        todo.pop_back();
        const uint32_t mask = rec[i].w0 & 0xffu, leaf_mask = (rec[i].w0 >> 8) & 0xffu;
        uint32_t child = rec[i].w1;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            if ((size_t)child >= rec.size()) return false;
            if ((leaf_mask >> ci) & 1u) {
                const uint32_t alpha = rec[child].w0 >> 24, refr = rec[child].w1 & 0xffu;
                if (alpha != 0u && (alpha != 255u || refr == 0u || refr == 85u)) return false;
            } else {
                todo.push_back(child);
            }
            ++child;
        }
    }
    return true;
}


}  // namespace vrt
