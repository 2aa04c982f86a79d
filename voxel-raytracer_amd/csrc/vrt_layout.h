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

}  // namespace vrt
