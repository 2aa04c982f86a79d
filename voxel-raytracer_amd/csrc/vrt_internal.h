// vrt_internal.h -- what the translation units of libvrt_hip.so share: the context behind `vrt_ctx`, the kernel variant
// table, error plumbing and the few helpers that cross files. Host code only; nothing here is exported.
//
//   vrt_scene.cpp      create / destroy, uniforms, camera, uploads, the layouts on the device (ensure_analysis)
//   vrt_dispatch.cpp   enqueue(): kernel arguments, variant choice, feedback scheduling, ray tables; the vrt_dispatch* entry points
//   vrt_display.cpp    the display pass and the fused frame call
//   vrt_patch.cpp      edits without re-upload: patch plan / apply / batches / compaction
//   vrt_raygen.cpp     per-projection ray-generation tables (pure host arithmetic)
//   vrt_launch_*.hip   the ONLY files that hold device code: kernel instantiations behind vrt_launch.h
//   vrt_multi.hip      several devices behind one handle (uses the public API of the per-device contexts)
#pragma once
#include "../../include/vrt.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "vrt_args.h"
#include "vrt_layout.h"

// Variant 0 is what the library ships: the v4 traversal, seven waves per SIMD (five for the full path tracer, which runs
// v4's general march loop); the dispatcher takes v3 when a primary / primary + shadow launch has its eye inside a medium,
// v2 when the scene has no wide form, v1 when it has a unit-size internal node. Variants 1, 4 and 20 select those fallbacks explicitly (tests). Everything else is A/B material and is only
// compiled into the library with `make AB=1` (-DVRT_AB_VARIANTS); vrt_set_variant() refuses what is not there.
#ifdef VRT_AB_VARIANTS
#define VRT_AB 1
#else
#define VRT_AB 0
#endif

struct Variant {
    int trav;          // 2: bit-indexed descent with restart anchors (vrt_kernels.hip.h); 1: baseline (vrt_kernels_v1.hip.h)
    bool use_lds;      // stage the level-order record prefix in LDS
    int tw;            // tile width in pixels (tile = tw x 64/tw)
    int block;         // threads per workgroup
    uint32_t lds_cap;  // max records staged in LDS
    int blocks_per_cu; // > 0: persistent grid of CUs*blocks_per_cu workgroups; 0: one pass over all tiles
    int wpe;           // waves per SIMD the register allocator is held to (1 = unconstrained)
};

inline constexpr Variant kVariants[] = {
    /*0*/ {4, false, 8, 64, 0, 0, 7},
    /*1*/ {1, false, 8, 256, 0, 0, 1},
    /*2*/ {2, true, 8, 256, 2048, 0, 1},
    /*3*/ {2, false, 16, 256, 0, 0, 1},
    /*4*/ {2, false, 8, 256, 0, 0, 1},
    /*5*/ {2, true, 8, 1024, 8192, 0, 1},
    /*6*/ {2, false, 8, 256, 0, 8, 1},
    /*7*/ {1, true, 8, 256, 2048, 0, 1},
    /*8*/ {2, false, 8, 64, 0, 0, 1},
    /*9*/ {2, false, 8, 128, 0, 0, 1},
    /*10*/ {2, false, 8, 256, 0, 0, 5},
    /*11*/ {2, false, 8, 256, 0, 0, 6},
    /*12*/ {2, false, 8, 256, 0, 0, 8},
    /*13*/ {3, false, 8, 256, 0, 0, 1},
    /*14*/ {3, false, 8, 256, 0, 0, 6},
    /*15*/ {3, false, 8, 256, 0, 0, 8},
    /*16*/ {3, false, 8, 64, 0, 0, 1},
    /*17*/ {3, false, 8, 64, 0, 0, 8},
    /*18*/ {3, false, 16, 256, 0, 0, 1},
    /*19*/ {3, false, 8, 64, 0, 0, 7},
    /*20*/ {3, false, 8, 64, 0, 0, 6},  // v3 as round 1 shipped it (six waves per SIMD; seven with the shadow march)
    /*21*/ {4, false, 8, 64, 0, 0, 6},
    /*22*/ {4, false, 8, 64, 0, 0, 7},  // == variant 0
    /*23*/ {4, false, 8, 64, 0, 0, 8},
    /*24*/ {4, false, 8, 64, 0, 0, 1},
};
inline constexpr bool kVariantShipped[] = {true, true, false, false, true, false, false, false, false, false, false, false, false,
                                    false, false, false, false, false, false, false, true, false, true, false, false};
inline constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
static_assert(sizeof(kVariantShipped) / sizeof(kVariantShipped[0]) == (size_t)kNumVariants, "one flag per variant");

// Feedback scheduling state of one launch shape on one stream. Launches that repeat a shape on a stream (the frames of
// a camera path) share it: every sched_period-th of them also records what each tile cost, tile_order_kernel turns that
// into a heaviest-first workgroup order on the same stream, and the launches that follow start their workgroups in
// that order. The order is a permutation whatever the costs are, so a stale one (camera moved, scene edited) only
// loses speed, never pixels; states are per stream because the order buffer is rewritten in stream order.
struct SchedState {
    hipStream_t stream = nullptr;
    int width = 0, n_rows = 0, row0 = 0, row_stride = 0, tile_rows = 0, mode = 0;
    uint32_t n_tiles = 0, n_groups = 0;
    uint32_t *d_cost = nullptr, *d_order = nullptr;
    bool valid = false;        // d_order holds an order
    float cam[6] = {0, 0, 0, 0, 0, 0};  // eye and viewing direction of the launch the order was measured on (trace states)
    uint64_t launches = 0;
    uint64_t last_use = 0;
};

struct vrt_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    uint2 *d_nodes = nullptr;
    size_t nodes_capacity = 0;
    bool have_scene = false;
    bool have_camera = false;
    vrt_scene_info info{};
    vrt_params params{};
    float inv_proj[16]{}, inv_view[16]{}, cam_pos[4]{};
    int variant = 0;
    int denoise_variant = 0;  // VRT_OPT_DISPLAY_KERNEL: 0 two pixels per lane, each wave the cheaper walk; 1 one pixel per lane (A/B builds); 2, 3: one walk forced
    // scratch outputs for the host-buffer dispatch
    void *d_rgba = nullptr;
    void *d_id = nullptr;
    void *d_shown = nullptr;
    size_t scratch_pixels = 0;
    // edits collected between vrt_patch_begin and vrt_patch_end: applied to the host structures at once, sent to the
    // device together
    struct PatchBatch {
        bool open = false, dirty = false;
        size_t records_before = 0, cells_before = 0;
        std::vector<uint32_t> rewritten_records;
        std::vector<size_t> repointed_cells;
        bool roots_changed = false, wide_invalid = false;
        long texel_delta = 0;
    };
    PatchBatch batch;
    // vrt_dispatch_async: two lanes, each a stream + device images + "the copies have landed" event
    struct AsyncLane {
        hipStream_t stream = nullptr;
        void *d_rgba = nullptr, *d_id = nullptr;
        size_t pixels = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    };
    AsyncLane lane[2];
    int next_lane = 0;
    // optional per-launch hipEvent pairs (vrt_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> prof_events;  // 2 per slot
    size_t prof_count = 0;
    size_t prof_seen = 0, prof_stride = 1;  // every prof_stride-th launch is bracketed
    // host copy of the records: lets the dispatcher check the bit-indexed traversal's precondition
    // against the CURRENT world bounds (they arrive separately, through vrt_set_params)
    std::vector<vrt::Record> host_records;
    size_t uploaded_records = 0;  // size of host_records after the last full upload (patches append to it)
    size_t stream_texels = 0;     // texels of the reference's stream for the current tree (kept current by patches)
    bool dim_from_texels = false; // the uploaded tex_dim was ceil(cbrt(texels)): patches keep it that way
    bool analysis_valid = false;
    bool unit_internal = false;
    // wide layout (vrt_layout.h), rebuilt whenever the tree or the world bounds change
    bool wide_ok = false;
    vrt::WideTree wide;
    uint2 *d_cells = nullptr;     // cells_capacity cells in the layout of vrt_layout.h, then as many in the v4 form (cells4)
    uint32_t *d_roots = nullptr;  // 16 words: record and wide node of each wide root (vrt_common.hip.h KArgs::root_table)
    size_t cells_capacity = 0;
    // the full path tracer as two kernels (vrt_bounce.hip.h): deferred-bounce queues, sized for the largest launch so far
    bool full_split = false;
    int bounce_refill_below = 40, bounce_waves_per_simd = 6;   // vrt_ab_set_bounce (A/B builds; tools sweep them)
    struct DeferQueues {                  // one set per stream: launches on different streams may overlap
        hipStream_t stream = nullptr;
        float *rec = nullptr;
        uint32_t *count = nullptr;        // 2 * kDeferQueues counters, kDeferStride words apart: records written, records handed out
        size_t cap = 0;                   // records per queue
        uint64_t last_use = 0;
    };
    std::vector<DeferQueues> defer;
    uint64_t defer_tick = 0;
    // feedback scheduling of the default kernel (see SchedState)
    int sched_period = 16;                   // every n-th launch of a shape measures its tiles; 0 = off
    std::vector<SchedState> sched;
    uint64_t sched_tick = 0;
    bool order_lds_raised = false;            // tile_order_kernel's dynamic-LDS ceiling raised on THIS context's device
    // ray-generation tables, one per (inverse projection, width, height) seen lately (ray_table() below)
    struct RayTable {
        float inv_proj[16]{};
        int width = 0, height = 0;
        bool ok = false;            // the projection has the separable shape and the tables are on the device
        float z = 0.0f;
        float *d_tab = nullptr;     // width floats (x per column) then height floats (y per row)
        size_t capacity = 0;        // floats
        uint64_t last_use = 0;
    };
    std::vector<RayTable> ray_tables;
    uint64_t ray_tick = 0;
    // VRT_MODE_FULL as two passes (vrt_launch.h trace_full_two_pass): the option, what the uploaded tree allows, the seed buffers
    bool two_pass_on = true;                     // vrt_set_option(VRT_OPT_FULL_OPAQUE)
    bool heavy_split_on = true;                  // KArgs::split_count (VRT_OPT_HEAVY_TILES)
    int two_pass_form = 6;                       // 6 (default): both stages in one kernel built for six waves per SIMD; 5, 7: for five, seven; 1: two kernels
    bool scene_opaque = false;                   // every leaf has alpha 0, or alpha 255 and a refraction byte that is a surface (not 0 / 85)
    bool scene_opaque_valid = false;
    struct SeedBuffer { hipStream_t stream = nullptr; uint32_t *d = nullptr; size_t tiles = 0; uint64_t last_use = 0; };
    std::vector<SeedBuffer> seeds;               // one per stream: launches on different streams may overlap
    uint64_t seed_tick = 0;
    bool tight_root_on = true;                   // VRT_OPT_EMPTY_OCTANTS 2 = on without the tighter root
    bool root0_only_on = true;                   // VRT_OPT_EMPTY_OCTANTS 0: never tell the kernels that the world is empty outside wide root 0
    bool ray_tables_on = true;                   // VRT_OPT_RAY_TABLES 0: always the shader's own prologue (A/B, tests)
    const uint32_t *dbg_group_order = nullptr;  // vrt_set_tile_order: caller-owned buffers instead of the scheduler's
    uint32_t *dbg_tile_cost = nullptr;
    bool dbg_sched = false;
    std::string err;
};

inline int vrt_fail(vrt_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define VRT_HIP(c, call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return vrt_fail((c), VRT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace vrt_internal {

// vrt_scene.cpp
int ensure_analysis(vrt_ctx *c);     // (re)derives what depends on the world bounds: wide layout on the device, root table
int reserve_cells(vrt_ctx *c, size_t n_cells);
int upload_cells(vrt_ctx *c, size_t from, size_t n);
int upload_roots(vrt_ctx *c);
uint32_t dim_of_texels(size_t texels);   // src/main.cpp:266-268
int check_frame(vrt_ctx *c, int width, int height);
int ensure_scratch(vrt_ctx *c, size_t px);   // device images behind the host-buffer entry points

// vrt_dispatch.cpp
int enqueue(vrt_ctx *c, int width, int height, int row0, int n_rows, int tile_rows, int row_stride, int compact, int mode,
            void *d_rgba, void *d_id, hipStream_t s, const vrt_view *views = nullptr, int n_views = 1);
SchedState *sched_state(vrt_ctx *c, hipStream_t s, int width, int n_rows, int row0, int row_stride, int tile_rows, int mode,
                        uint32_t n_tiles, uint32_t n_groups);
bool measuring_launch(uint64_t launches, int period);
int launch_order_kernel(vrt_ctx *c, SchedState *st, hipStream_t s);
constexpr long kSchedMinGroups = 768;    // an eighth of a 1080p frame (1,013 groups) still gains 4 %; below, the 9 us order kernel costs more
constexpr long kSchedMaxGroups = 36864;  // tile_order_kernel keeps one word per group in LDS (144 KiB of 160)
constexpr size_t kSchedMaxStates = 16;
constexpr int kSchedDenoise = 100;              // SchedState::mode of the display pass
constexpr long kSchedMinDenoiseGroups = 256;    // two workgroups fit a CU: 1,024 tiles are two rounds
constexpr long kSchedMaxDenoiseGroups = 2048;   // beyond ~8,000 tiles (16 rounds) the tail is small and heaviest-first starts cost the halo reads their L2 locality: 4K nature 0.147 ms row-major, 0.157 ordered

// vrt_raygen.cpp
bool build_ray_table(const float *m, int W, int H, std::vector<float> &tab, float &z_out);
bool view_matrix_in_range(const float *m);

}  // namespace vrt_internal
