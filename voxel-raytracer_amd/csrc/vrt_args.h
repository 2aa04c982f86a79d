// vrt_args.h -- what the dispatcher (host) and the kernels (device) share by value: the kernel argument blocks and the
// constants both sides size buffers with. No device code: host translation units include this alone.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrt {

struct F3 { float x, y, z; };
struct I3 { int x, y, z; };

// One camera and the two images it renders into. A launch carries up to kMaxViews of them (blockIdx.y selects
// the view): frames of one scene that are known together -- a stereo pair, the next frames of a camera path, the
// views of a light-field rig -- share one launch, so the drain of one view's last waves is filled by the next
// view's first instead of idling the chip between launches.
constexpr int kMaxViews = 4;
constexpr int kGroupTiles = 4;  // tiles per scheduling group: 4 adjacent 8x8 tiles, i.e. 32 x 8 pixels
// The general full path tracer starts the heaviest groups of an ordered launch as kSplitParts waves per tile (one 8-pixel row each):
// at most kSplitMaxGroups groups (KArgs::split_count, vrt_sched.hip.h)
constexpr int kSplitParts = 8, kSplitMaxGroups = 64;   // (4 and 16 parts measured: room 1080p 0.93 / unstable against 0.86 at 8)
struct View {
    float inv_proj[16];
    float inv_view[16];
    float cam_pos[4];
    uint32_t *out_rgba;       // packed R | G<<8 | B<<16 | A<<24
    int2 *out_id;             // (voxelID, dist)
    uint32_t eye0, eye1;      // raw leaf words of the node that holds the eye (comp:445-449), looked up by the host
    // the primary rays' first lookup (at the eye), made by the host for the wide kernels (vrt_layout.h first_find)
    uint32_t first_w0, first_w1, first_node, first_anode;
    int first_s, first_as, first_valid;
    // Ray generation with its per-column and per-row parts made once per projection by the dispatcher (ray_table() in
    // vrt_dispatch.cpp, vrt_raygen.cpp): when the inverse projection has the shape every perspective or orthographic matrix gives it -- x
    // depends on the column only, y on the row only, z and w on neither -- gen_x[px], gen_y[py], gen_z hold view.xyz / w
    // of comp:630-634 (same float operations, made on the host), and gen_fast says that they do and that every
    // normalisation of the prologue stays inside the range where 1/x and sqrt need no range scaling (primary_ray_dir()).
    const float *gen_x, *gen_y;
    float gen_z;
    uint32_t gen_fast;
};

// Kernel arguments: passed by value (kernarg segment -> scalar loads, wave-uniform).
// The views of a launch: a kernel argument of its own. Indexed by blockIdx.y inside KArgs it made the compiler
// treat every argument as dynamically addressed and keep them live (88 instead of 69 VGPRs on gfx950, through SGPR
// spills into vector lanes); on its own it costs nothing.
struct ViewSet {
    View v[kMaxViews];
};

struct KArgs {
    int n_views;              // gridDim.y
    float voxel_scale;
    int wmin[3];
    int wmax[3];
    float global_light[4];
    float light_dir[3];
    // the shadow ray's set-up (comp:335-345), the same for every ray of a launch: made by the dispatcher from light_dir
    // with the shader's operations -- 1/d or 1e20, sign * 1e-3, d > 0 -- instead of 70 vector instructions per wave
    float light_inv[3], light_push[3], light_dposf[3];
    int light_dpos[3];
    int shade_fast;           // globalLight and lightDir are finite and at most 2^30: the shading quotients x / PI are in range (div_pi_inrange())
    int highlighted[3];
    int tex_dim;
    int width, height;
    // rows traced by this launch: local row j in [0, n_rows) maps to frame row
    //   y = row0 + (j / tile_rows) * row_stride + (j % tile_rows)
    int row0, n_rows, tile_rows, row_stride;
    int compact;              // 1: outputs indexed by local row j, 0: by frame row y
    // the two index divisions of a wave's prologue, prepared by the host (enqueue() in vrt_dispatch.cpp):
    uint32_t tiles_x_magic;   // floor(2^32 / tiles_x) + 1 when tile / tiles_x == umulhi(tile, magic) for every tile, else 0
    int row_mode;             // 1: one row tile (y = row0 + j); 2: tile_rows == 8 == tile height (y = row0 + ty * row_stride + ly); 0: divide
    const uint2 *nodes;       // level-ordered records (vrt_layout.h), root = record 0
    uint32_t n_records;
    uint32_t lds_records;     // prefix of `nodes` staged in LDS by each workgroup
    // wide layout (vrt_layout.h): 64 cells per node; roots = octree records where a wide tree starts
    const uint2 *cells;
    const uint2 *cells4;      // the same cells in the form of the v4 kernels (vrt_layout.h to_cell4)
    uint32_t n_roots;
    // wide roots: [0..7] octree record of each root, [8..15] its wide node (device memory; read only by the record
    // walk that a lookup outside wide root 0 takes). Root 0's node and log2 side also travel by value.
    const uint32_t *root_table;
    uint32_t root0_node;
    int root0_shift;
    int root0_min[3];         // minimum corner of wide root 0's cube (valid when n_roots > 0)
    int root0_only;           // 1: every record outside wide root 0's subtree is an absent child -- the world is empty outside that cube
    // Feedback scheduling (SCHED flavours of trace_kernel; vrt_dispatch.cpp owns the buffers). The unit is a GROUP of
    // kGroupTiles consecutive tiles. bit 0: the g-th group of tiles the launch starts is group_order[g] (a permutation
    // of the launch's groups, heaviest first). bit 1: every wave leaves the clock ticks its tile took in
    // tile_cost[tile], from which tile_order_kernel derives the next order.
    const uint32_t *group_order;
    uint32_t *tile_cost;
    // MODE 2 of trace_kernel, launches under an order: *split_count = how many groups at the head of group_order are
    // traced as kSplitParts waves per tile (the order kernel counts the groups above 3/4 of the heaviest one's ticks, at most
    // kSplitMaxGroups; 0 when the heaviest tile does not outlast its even share of the frame -- no tail to shorten). The grid then holds
    // kSplitMaxGroups * kGroupTiles * (kSplitParts - 1) workgroups more than tiles; the ones no group needs leave at once. null: none.
    const uint32_t *split_count;
    // Deferred diffuse bounces of the full path tracer (MODE 3 of trace_kernel, vrt_bounce.hip.h): kDeferQueues queues of
    // defer_cap ray records each, structure of arrays (plane p of queue q starts at defer_rec + (p * kDeferQueues + q) * defer_cap),
    // defer_count[q * kDeferStride] = records in queue q, defer_count[(kDeferQueues + q) * kDeferStride] = records already
    // handed out by bounce_kernel: every counter in a cache line of its own (atomics on one line are served one at a time).
    float *defer_rec;
    uint32_t *defer_count;
    uint32_t defer_cap;
};
constexpr uint32_t kSeedPlanesHost = 5;  // words per pixel of the two-pass full path tracer's seed (vrt_common.hip.h kSeedPlanes)
constexpr uint32_t kDeferQueues = 64;   // a wave appends to queue (tile % 64): sixty-four counters share the atomic traffic
constexpr uint32_t kDeferStride = 64;   // words between two counters: 256 bytes
constexpr uint32_t kDeferPlanes = 19;   // o[3] d[3] tint[3] fc[3] iof weight mc[3] md out_offset

namespace v3 { constexpr int kAnchorShift = 6; }  // restart point of the wide traversals: the wide node of side 64 the ray is in

}  // namespace vrt
