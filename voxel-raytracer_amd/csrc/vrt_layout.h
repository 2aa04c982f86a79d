// vrt_layout.h -- host-side conversion of the reference's flattened octree
// (the RGBA8UI texel stream of src/octree.cpp:573-682, format in SURVEY.md
// Appendix A) into the device layout the gfx950 kernels read.
//
// Device layout: one array of 8-byte records in LEVEL ORDER (breadth first),
// root = record 0, the children of a node stored contiguously in child-index
// order (absent children skipped):
//
//   internal record : w0 = child_mask | leaf_mask << 8     (bit i = child i present / is a leaf)
//                     w1 = index of the first child record
//   leaf record     : w0 = R | G << 8 | B << 16 | alpha << 24   (texel0.rgb, texel1.a)
//                     w1 = refr | illum << 8 | k << 16          (texel1.rgb)
//
// Why: the texel stream needs two DEPENDENT 4-byte fetches per tree level
// (header, then pointer) plus two for a leaf; one record fetch per level carries
// the same information, and level order makes the hot top of the tree a prefix
// of the array that a workgroup can stage in LDS with one coalesced copy.
// octreeFind's result is a function of the query point only, so the layout is
// invisible in the outputs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace vrt {

struct Record { uint32_t w0, w1; };

struct Layout {
    std::vector<Record> records;
    std::vector<uint32_t> level_start;  // first record index of each depth (+ end sentinel)
    uint32_t n_internal = 0, n_leaves = 0, max_depth = 0;
};

// Returns false (and sets err) when the stream cannot be walked within limits.
// An empty stream yields a single root record with no children.
bool build_layout(const uint8_t *texels, size_t used_bytes, Layout &out, std::string &err);

// True when some INTERNAL node's AABB (under the world bounds given, split as the shader splits:
// mid = min + (max - min) / 2) is no larger than one voxel in every axis. octree_texture() never
// writes such a node; the bit-indexed traversal (vrt_kernels.hip.h) does not support it and the
// dispatcher then falls back to the explicit-AABB kernels.
bool has_unit_internal_node(const std::vector<Record> &records, const int wmin[3], const int wmax[3]);

// ---------------------------------------------------------------------------------------------
// Wide layout ("64-trees"), built from the record array once the world bounds are known.
//
// Inside an octree node whose AABB is an aligned cube of side 2^S (S even, >= 2) two octree levels
// are collapsed into one DENSE node of 4x4x4 = 64 cells of side 2^(S-2), indexed directly by two
// bits of each coordinate (cell = x2 << 4 | y2 << 2 | z2) -- no mask, no popcount, one 8-byte load
// per TWO octree levels. A cell holds everything octreeFind would return for any point in it:
//
//   cell.w0, cell.w1[23:0] : the leaf words (0/0 = empty space); the refraction byte w1[7:0] is stored as 0
//                            when the alpha byte w0[31:24] is 0 (see leaf_props() in vrt_layout.cpp)
//   cell.w1[28:24]         : t = log2(side) of the octree node the point falls in (leaf or absent
//                            child; it may be larger than the cell, then the cell is one of several
//                            copies), from which the kernel rebuilds that node's AABB
//   cell.w1[31] = 1        : the cell is subdivided further: w0 = index of the child wide node
//
// Wide trees are rooted at the octree nodes first met on the way down from the root that are
// aligned cubes of even log2 side ("roots"); the few levels above them (the reference's world
// [-1023,1024)^3 is not a power of two) stay in the record array and are walked with explicit AABBs.
struct WideCell { uint32_t w0, w1; };
struct WideRoot { uint32_t record; uint32_t node; int shift; int origin[3]; };  // origin: minimum corner of the root's cube
struct WideTree {
    std::vector<WideCell> cells;   // 64 per wide node
    std::vector<WideRoot> roots;
    std::vector<uint32_t> node_record;  // per wide node: the record of its octree node (patches reuse unchanged nodes)
    uint32_t n_nodes = 0;
};
constexpr uint32_t kWideInternal = 0x80000000u;
constexpr int kMaxWideRoots = 8;

// The same cell as the v4 kernels read it (vrt_kernels_v4.hip.h), chosen so that the march loop needs the cheapest
// instructions gfx950 has for what it does with a cell (profiles/r02_valu_rate.txt):
//   subdivided : w0 = BYTE offset of the child wide node (node * 512), w1 = 0
//   leaf/empty : w0 = the leaf word 0 (R | G<<8 | B<<16 | alpha<<24), unchanged
//                w1[ 7: 0] = medium byte: the refraction byte, with 85 (refraction 1.0, what the shader substitutes for
//                            empty space, comp:318-326) where the layout above stores 0 -- two media differ iff these differ
//                w1[15: 8] = illumination byte, w1[22:16] = k bits 6..0, w1[28] = k bit 7
//                w1[27:23] = t + 1 (t = log2 side of the octree node found): (w1 & 0x0f800000) + 0x3f000000 is the float
//                            2^t and 0x40000000 - (w1 & 0x0f800000) the float 2^-t, so the node's planes are four plain
//                            f32 operations per axis on floor(p) instead of integer shifts and conversions
//                w1[29]    = the refraction byte was 0 although alpha > 0 (restores word 1 exactly for shading)
// t + 1 >= 1 tells a leaf/empty cell from a subdivided one.
inline WideCell to_cell4(WideCell c) {
    if (c.w1 & kWideInternal) return WideCell{c.w0 << 9, 0u};
    const uint32_t t = (c.w1 >> 24) & 31u, b = c.w1 & 0xffu, illum = (c.w1 >> 8) & 0xffu, k = (c.w1 >> 16) & 0xffu;
    const uint32_t raw0 = (b == 0u && (c.w0 >> 24) != 0u) ? 1u : 0u;
    return WideCell{c.w0, (b ? b : 85u) | (illum << 8) | ((k & 0x7fu) << 16) | ((t + 1u) << 23) | ((k >> 7) << 28) | (raw0 << 29)};
}

// True when the tree has nothing outside wide root 0: there is one wide root, and from the octree root down to that root's
// record every node has exactly one child, an internal one (everything else is absent children, i.e. empty space).
bool content_only_in_root0(const std::vector<Record> &records, const WideTree &wt);

// With content_only_in_root0(): the deepest wide node that can stand in for wide root 0 in a launch whose eyes are `eyes`
// (n of them, floor of the camera positions): as long as all of a node's content sits in ONE of its 64 cells, that cell is
// subdivided, its cube holds every eye and its side stays >= 2^min_shift, the child takes over. in/out: node, shift, origin.
void tighten_root0(const WideTree &wt, const int (*eyes)[3], int n, int min_shift, uint32_t &node, int &shift, int origin[3]);

// True when pathTrace (raytracing.comp:435-622) cannot take its translucent branch (:546-572) or absorb (:482-486, 512-516) anywhere in
// this tree for an eye in empty space: every leaf has alpha 0 (never a hit: its medium byte reads as empty space) or alpha 255 with a
// refraction byte that makes it a surface (not 0 and not 85, which the hit test cannot tell from empty space). The dispatcher then
// runs VRT_MODE_FULL without a ray stack (vrt_full.hip.h bounce_pixel).
bool tree_is_opaque(const std::vector<Record> &records);

// Returns false when the scene cannot be expressed (an internal node of unit size inside an aligned
// cube, or more than kMaxWideRoots roots): the dispatcher then uses the record-array kernels.
bool build_wide(const std::vector<Record> &records, const int wmin[3], const int wmax[3], WideTree &out, std::string &why);

// Host mirror of the device lookup (tests): the octreeFind result for point p through the wide layout.
// Returns 1 and the leaf words for a leaf, 0 for empty space; mn/mx receive the node AABB.
int wide_find_host(const std::vector<Record> &records, const WideTree &wt, const int wmin[3], const int wmax[3],
                   const int p[3], uint32_t &w0, uint32_t &w1, int mn[3], int mx[3]);

// The raw leaf words of the octree node that holds point p (0/0 for empty space or a point outside the world): the
// lookup the shader makes once per ray at the eye (raytracing.comp:445-449). The eye is the same for every ray of a
// view, so the dispatcher makes it once on the host and hands the two words to the kernel.
void eye_lookup(const std::vector<Record> &records, const int wmin[3], const int wmax[3], const int p[3], uint32_t &w0,
                uint32_t &w1);

// The first lookup of every primary ray of a view is made at the eye too: the same point, hence the same wide-layout
// walk, for every ray. first_find() makes it once on the host in the form the wide kernels keep it (the cell found
// and the restart state: current wide node, anchor node, their log2 sides). False when the wide kernels would not
// start at wide root 0 (no wide layout, eye outside the world or outside that root's cube): they then look it up.
struct FirstFind { uint32_t w0, w1, node, anode; int s, as; };
bool first_find(const WideTree &wide, const int wmin[3], const int wmax[3], const int p[3], int anchor_shift, FirstFind &out);

// ---------------------------------------------------------------------------------------------
// Edits without a rebuild. A voxel edit changes the octree below some ancestor A of the voxel and
// nothing else. When A is an INTERNAL node before and after the edit, the device structures can be
// patched: the caller re-emits A's sub-tree as records (level order, A = record 0, child indices local
// to the sub-tree); the new sub-tree is compared with the old one and only the child blocks that differ --
// normally the ones along the path to the edited voxel -- are APPENDED to the array (children still come
// after their parents; unchanged sub-trees keep their records and are shared), A's own record is rewritten in
// place, and -- when A is the octree node of a wide node -- the wide nodes along the same path are rebuilt
// at the end of the cell array (wide nodes over unchanged records are reused) and the one cell (or root
// table entry) that referred to A's old wide node is repointed. Replaced blocks and wide nodes stay behind as
// unreferenced garbage until the next full upload.
struct PatchSite {
    int depth = 0;                 // of A below the root (>= 1)
    uint8_t path[16] = {};         // child index taken at each level, root first
    uint32_t record = 0;           // A's record
    int shift = -1;                // log2 side of A when A is the octree node of a wide node, else -1
    int root_index = -1;           // A is wide root `root_index`, or
    uint32_t parent_node = 0;      //   cell `parent_cell` of wide node `parent_node` points at A's wide node
    uint32_t parent_cell = 0;
};

// The deepest ancestor of voxel p, at depth 1..max_depth, that is an internal node and (when the wide layout is in
// use) the octree node of a wide node. False when there is none: the edit needs a full upload.
bool plan_patch(const std::vector<Record> &records, const WideTree &wide, bool wide_in_use, const int wmin[3],
                const int wmax[3], const int p[3], int max_depth, PatchSite &site);

// A record {kKeep, kKeep} in an internal child's place of an emitted sub-tree means "this child and everything below
// it is unchanged": the emitter then only has to walk the nodes that contain the edited voxel.
constexpr uint32_t kKeep = 0xffffffffu;

struct PatchRanges {               // what changed, for the device copies
    size_t records_appended_from = 0;   // records [from, size) are new
    size_t cells_appended_from = 0;     // cells [from, size) are new
    bool cell_repointed = false;        // cells[parent_node * 64 + parent_cell] changed
    bool wide_invalid = false;          // the wide layout could not be patched: rebuild it from the records
    long texel_delta = 0;               // change of the reference stream's texel count (stream_texels) by this patch
};

// Texels the sub-tree under internal record `top` occupies in the reference's stream (one header per internal node,
// one pointer per present child, two texels per leaf): u_texDim, which the shader folds into
// the voxel ids it writes, is ceil(cbrt()) of the stream's total, so a patch has to keep that total current.
size_t stream_texels(const Record *records, size_t n_records, uint32_t top);

// The sub-tree under the node reached from the root by `depth` child indices, re-indexed from 0 (what the host
// library emits from the pointer octree). False when the path leaves the tree or does not end at an internal node.
// With `toward` (and the world bounds) only the nodes that contain that voxel are expanded; every other internal
// child is emitted as a kKeep record.
bool extract_subtree(const std::vector<Record> &records, const uint8_t *path, int depth, std::vector<Record> &sub,
                     const int *toward = nullptr, const int *wmin = nullptr, const int *wmax = nullptr);

// Drops what patches left behind: the records reachable from the root, re-laid in level order (children of a node
// contiguous, after their parent -- the invariants of build_layout), everything else gone. The wide layout built over the
// old indices is void afterwards.
void compact_records(std::vector<Record> &records);

// sub: A's new sub-tree, n_sub records, sub[0] = A as an internal record (its child mask may be empty). False (nothing
// modified) when sub is malformed.
bool apply_patch(std::vector<Record> &records, WideTree &wide, bool wide_in_use, const PatchSite &site, const Record *sub,
                 size_t n_sub, PatchRanges &out, std::string &why);

}  // namespace vrt
