// vrt_capi.hip -- the C-ABI of include/vrt.h over the gfx950 kernels.
// Built by hipcc --offload-arch=gfx950 into libvrt_hip.so. No CPU fallback.
#include "../../include/vrt.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "vrt_kernels.hip.h"
#include "vrt_kernels_v1.hip.h"
#include "vrt_kernels_wide.hip.h"
#include "vrt_kernels_v4.hip.h"
#include "vrt_full.hip.h"
#ifdef VRT_AB_VARIANTS
#include "vrt_bounce.hip.h"   // the two-kernel full path tracer: an experiment that lost (profiles/r02_b_*), kept for A/B builds
#endif
#include "vrt_denoise.hip.h"
#include "vrt_layout.h"

namespace {

thread_local std::string g_create_error;

struct Variant {
    int trav;          // 2: bit-indexed descent with restart anchors (vrt_kernels.hip.h); 1: baseline (vrt_kernels_v1.hip.h)
    bool use_lds;      // stage the level-order record prefix in LDS
    int tw;            // tile width in pixels (tile = tw x 64/tw)
    int block;         // threads per workgroup
    uint32_t lds_cap;  // max records staged in LDS
    int blocks_per_cu; // > 0: persistent grid of CUs*blocks_per_cu workgroups; 0: one pass over all tiles
    int wpe;           // waves per SIMD the register allocator is held to (1 = unconstrained)
};

// Variant 0 is what the library ships: the v4 traversal, seven waves per SIMD (five for the full path tracer, which runs
// v4's general march loop); the dispatcher takes v3 when a primary / primary + shadow launch has its eye inside a medium,
// v2 when the scene has no wide form, v1 when it has a unit-size internal node. Variants 1, 4 and 20 select those fallbacks explicitly (tests). Everything else is A/B material and is only
// compiled into the library with `make AB=1` (-DVRT_AB_VARIANTS); vrt_set_variant() refuses what is not there.
#ifdef VRT_AB_VARIANTS
#define VRT_AB 1
#else
#define VRT_AB 0
#endif
const Variant kVariants[] = {
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
constexpr bool kVariantShipped[] = {true, true, false, false, true, false, false, false, false, false, false, false, false,
                                    false, false, false, false, false, false, false, true, false, true, false, false};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
static_assert(sizeof(kVariantShipped) / sizeof(kVariantShipped[0]) == (size_t)kNumVariants, "one flag per variant");

}  // namespace

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
    int denoise_variant = 0;  // pixels per lane: 0 -> two, 1 -> one (vrt_debug_set_denoise_variant)
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
    int bounce_refill_below = 40, bounce_waves_per_simd = 6;   // vrt_debug_set_bounce (tools sweep them)
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
    bool tight_root_on = true;                   // vrt_debug_set_root0_only(2 = on without the tighter root)
    bool root0_only_on = true;                   // vrt_debug_set_root0_only(0): never tell the kernels that the world is empty outside wide root 0
    bool ray_tables_on = true;                   // vrt_debug_set_ray_tables(0): always the shader's own prologue (A/B, tests)
    const uint32_t *dbg_group_order = nullptr;  // vrt_debug_set_tile_order: caller-owned buffers instead of the scheduler's
    uint32_t *dbg_tile_cost = nullptr;
    bool dbg_sched = false;
    std::string err;
};

namespace {

int fail(vrt_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define VRT_HIP(c, call)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail((c), VRT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <int MODE, class TRAV, int TW, int BLOCK, int WPE, bool PERSIST = false, int SCHED = 0>
// ev0/ev1 (both or neither): events attached to THIS dispatch packet (hipExtLaunchKernel), so their elapsed time is
// the kernel's own begin-to-end time, as a profiler reports it, without the latency of separate event markers
hipError_t launch_one(const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds_bytes, hipStream_t s,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr) {  // grid.y = a.n_views
    void (*kernel)(const vrt::KArgs, const vrt::ViewSet) = &vrt::trace_kernel<MODE, TRAV, TW, BLOCK, WPE, PERSIST, SCHED>;
    if (lds_bytes > 48 * 1024) {  // above the default dynamic-LDS ceiling: opt in (CDNA4 has 160 KiB per CU). The attribute
        // belongs to the (function, device) pair, so it is set on every such launch (LDS-staging A/B variants only)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
    }
    if (ev0 || ev1)   // either may be null: the two-kernel full path tracer times from the first kernel's start to the second one's end
        hipExtLaunchKernelGGL(kernel, dim3(grid, a.n_views), dim3(BLOCK), lds_bytes, s, ev0, ev1, 0, a, vs);
    else
        hipLaunchKernelGGL(kernel, dim3(grid, a.n_views), dim3(BLOCK), lds_bytes, s, a, vs);
    return hipGetLastError();
}

// The combinations that exist in the feedback-scheduled flavours too (KArgs::group_order / tile_cost choose one).
template <int MODE, class TRAV, int TW, int BLOCK, int WPE>
hipError_t launch_sched(const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    switch ((a.group_order ? 1 : 0) | (a.tile_cost ? 2 : 0)) {
        case 1: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 2>(a, vs, grid, lds, s, ev0, ev1);
        case 3: return launch_one<MODE, TRAV, TW, BLOCK, WPE, false, 3>(a, vs, grid, lds, s, ev0, ev1);
        default: return launch_one<MODE, TRAV, TW, BLOCK, WPE>(a, vs, grid, lds, s, ev0, ev1);
    }
}

// The instantiated (traversal, LDS prefix, tile width, workgroup size, waves-per-SIMD) combinations.
template <int MODE>
hipError_t launch_mode(const Variant &v, const vrt::KArgs &a, const vrt::ViewSet &vs, int grid, size_t lds, hipStream_t s,
                       hipEvent_t ev0, hipEvent_t ev1) {
    using V1 = vrt::v1::Trav<false>;
    using V2 = vrt::v2::Trav<false>;
#if VRT_AB
    using V1L = vrt::v1::Trav<true>;
    using V2L = vrt::v2::Trav<true>;
#endif
    using V3 = vrt::v3::Trav;
    using V4 = vrt::v4::Trav;
#if VRT_AB
    if (v.blocks_per_cu > 0) {  // the persistent (grid-stride) form exists for one combination
        if (v.trav == 2 && !v.use_lds && v.tw == 8 && v.block == 256 && v.wpe == 1)
            return launch_one<MODE, V2, 8, 256, 1, true>(a, vs, grid, lds, s, ev0, ev1);
        return hipErrorInvalidValue;
    }
#endif
    const int key = v.trav * 1000000 + (v.use_lds ? 100000 : 0) + v.tw * 1000 + (v.block / 64) * 10 + v.wpe;
    switch (key) {
        // shipped: the default (v4) and the three fallbacks the dispatcher may take
        case 4000000 + 8000 + 10 + 7: return launch_sched<MODE, V4, 8, 64, 7>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 6: return launch_sched<MODE, V3, 8, 64, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 7: return launch_sched<MODE, V3, 8, 64, 7>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 1: return launch_one<MODE, V2, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 1000000 + 8000 + 40 + 1: return launch_one<MODE, V1, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
#if VRT_AB
        case 1100000 + 8000 + 40 + 1: return launch_one<MODE, V1L, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 5: return launch_one<MODE, V2, 8, 256, 5>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 6: return launch_one<MODE, V2, 8, 256, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 40 + 8: return launch_one<MODE, V2, 8, 256, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 16000 + 40 + 1: return launch_one<MODE, V2, 16, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 80 + 1: return launch_one<MODE, V2, 8, 512, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 10 + 1: return launch_one<MODE, V2, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2000000 + 8000 + 20 + 1: return launch_one<MODE, V2, 8, 128, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2100000 + 8000 + 40 + 1: return launch_one<MODE, V2L, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 2100000 + 8000 + 160 + 1: return launch_one<MODE, V2L, 8, 1024, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 1: return launch_one<MODE, V3, 8, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 6: return launch_sched<MODE, V3, 8, 256, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 40 + 8: return launch_one<MODE, V3, 8, 256, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 1: return launch_one<MODE, V3, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 8000 + 10 + 8: return launch_one<MODE, V3, 8, 64, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 3000000 + 16000 + 40 + 1: return launch_one<MODE, V3, 16, 256, 1>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 6: return launch_sched<MODE, V4, 8, 64, 6>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 8: return launch_sched<MODE, V4, 8, 64, 8>(a, vs, grid, lds, s, ev0, ev1);
        case 4000000 + 8000 + 10 + 1: return launch_sched<MODE, V4, 8, 64, 1>(a, vs, grid, lds, s, ev0, ev1);
#endif
        default: return hipErrorInvalidValue;
    }
}

constexpr long kSchedMinGroups = 768;    // an eighth of a 1080p frame (1,013 groups) still gains 4 %; below, the 9 us order kernel costs more
constexpr long kSchedMaxGroups = 36864;  // tile_order_kernel keeps one word per group in LDS (144 KiB of 160)
constexpr size_t kSchedMaxStates = 16;
constexpr int kSchedDenoise = 100;              // SchedState::mode of the display pass
constexpr long kSchedMinDenoiseGroups = 256;    // two workgroups fit a CU: 1,024 tiles are two rounds

// The scheduling state for this launch shape on this stream (created on first use; the least recently used one is
// recycled when there are kSchedMaxStates). nullptr when device memory for it cannot be had: the launch then runs plain.
SchedState *sched_state(vrt_ctx *c, hipStream_t s, int width, int n_rows, int row0, int row_stride, int tile_rows, int mode,
                        uint32_t n_tiles, uint32_t n_groups) {
    uint64_t &tick = c->sched_tick;
    ++tick;
    for (SchedState &st : c->sched)
        if (st.stream == s && st.width == width && st.n_rows == n_rows && st.row0 == row0 && st.row_stride == row_stride &&
            st.tile_rows == tile_rows && st.mode == mode && st.n_tiles == n_tiles && st.n_groups == n_groups) {
            st.last_use = tick;
            return &st;
        }
    SchedState *slot = nullptr;
    if (c->sched.size() < kSchedMaxStates) {
        c->sched.emplace_back();
        slot = &c->sched.back();
    } else {
        for (SchedState &st : c->sched)
            if (!slot || st.last_use < slot->last_use) slot = &st;
        // the recycled buffers may still be read by launches in flight on the old stream
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;
        (void)hipFree(slot->d_cost);
        (void)hipFree(slot->d_order);
        *slot = SchedState{};
    }
    // whole groups of ticks; the words past the last tile are never written and must read as zero
    const size_t cost_bytes = (size_t)n_groups * vrt::kGroupTiles * sizeof(uint32_t);
    if (hipMalloc((void **)&slot->d_cost, cost_bytes) != hipSuccess ||
        hipMalloc((void **)&slot->d_order, (size_t)n_groups * sizeof(uint32_t)) != hipSuccess ||
        hipMemsetAsync(slot->d_cost, 0, cost_bytes, s) != hipSuccess) {
        (void)hipFree(slot->d_cost);
        (void)hipFree(slot->d_order);
        *slot = SchedState{};  // an empty state matches no launch and is the first to be recycled
        (void)hipGetLastError();
        return nullptr;
    }
    slot->stream = s; slot->width = width; slot->n_rows = n_rows; slot->row0 = row0; slot->row_stride = row_stride;
    slot->tile_rows = tile_rows; slot->mode = mode; slot->n_tiles = n_tiles; slot->n_groups = n_groups;
    slot->last_use = tick;
    return slot;
}

// An order measured from one pose says little about a frame from a very different one: re-measure at once, instead of
// waiting out the period, when the eye has moved by more than 16 world units or the view has turned by more than ~8
// degrees since the order was taken (a cut, a teleport; ordinary camera motion stays far below both per period).
bool camera_jumped(const float was[6], const float now[6]) {
    float d2 = 0.0f, dot = 0.0f, n0 = 0.0f, n1 = 0.0f;
    for (int k = 0; k < 3; ++k) {
        d2 += (now[k] - was[k]) * (now[k] - was[k]);
        dot += now[3 + k] * was[3 + k];
        n0 += was[3 + k] * was[3 + k];
        n1 += now[3 + k] * now[3 + k];
    }
    if (!(d2 <= 16.0f * 16.0f)) return true;           // also true for NaN
    return !(dot * dot >= 0.98f * n0 * n1 && dot >= 0.0f);  // cos(8 deg)^2 = 0.98
}

// Which launches of a shape record tile times: the second one (warm), then every period-th; period 1 = all of them.
static inline bool measuring_launch(uint64_t launches, int period) {
    return period == 1 || launches % (uint64_t)period == 1;
}

// After a measuring launch, on the same stream: reads that launch's ticks, rewrites the order the next launches read.
int launch_order_kernel(vrt_ctx *c, SchedState *st, hipStream_t s) {
    bool &raised = c->order_lds_raised;   // per context, i.e. per device: the attribute does not carry over to another one
    const size_t lds = (size_t)st->n_groups * sizeof(uint32_t);
    if (lds > 48 * 1024 && !raised) {
        VRT_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&vrt::tile_order_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kSchedMaxGroups * sizeof(uint32_t))));
        raised = true;
    }
    hipLaunchKernelGGL(vrt::tile_order_kernel, dim3(1), dim3(1024), lds, s, (const uint4 *)st->d_cost, st->n_groups, st->d_order);
    VRT_HIP(c, hipGetLastError());
    st->valid = true;
    return VRT_OK;
}

// ---- ray generation, the part that does not depend on the pixel as a whole ---------------------------------------------
// comp:626-634: u = px / W * 2 - 1, v likewise, view = invProjection * (u, v, -1, 1), view /= view.w. For the inverse of
// any perspective or orthographic projection x depends on u alone, y on v alone, z and w on neither, so the W + H + 1
// distinct values are computed here once per (matrix, W, H) -- by the SAME float operations in the same order, which is
// what makes them the same bits (this file is compiled with -ffp-contract=off like the kernels; x86-64 float arithmetic is
// IEEE binary32) -- and the kernels read them (View::gen_x/gen_y/gen_z) instead of making two integer->float conversions,
// five divisions and a 4x4 product per pixel. A zero term's SIGN can depend on the other coordinate ((+0) * v): each value
// is therefore evaluated with the other coordinate at +1 and at -1 and must come out the same bits, otherwise the table
// is refused and the kernels run the shader's own prologue. Returns false when the matrix is not of that shape.
bool build_ray_table(const float *m, int W, int H, std::vector<float> &tab, float &z_out) {
    static const int kZeros[10] = {4, 8, 12, 1, 9, 13, 2, 6, 3, 7};   // column-major m[c * 4 + r]
    for (int k : kZeros)
        if (!(m[k] == 0.0f)) return false;
    const auto row = [&](int r, float u, float v) {   // mat_vec() of vrt_common.hip.h with (x, y, z, w) = (u, v, -1, 1)
        return (m[0 * 4 + r] * u + m[1 * 4 + r] * v) + (m[2 * 4 + r] * -1.0f + m[3 * 4 + r] * 1.0f);
    };
    const auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
    const float w = row(3, 1.0f, 1.0f), z = row(2, 1.0f, 1.0f);
    for (int k = 1; k < 4; ++k) {
        const float u = (k & 1) ? -1.0f : 1.0f, v = (k & 2) ? -1.0f : 1.0f;
        if (!same(row(3, u, v), w) || !same(row(2, u, v), z)) return false;
    }
    if (!(w == w) || !(z == z)) return false;
    const bool divide = fabsf(w) > 1e-6f;
    tab.resize((size_t)W + (size_t)H);
    double hi = 0.0;
    for (int px = 0; px < W; ++px) {
        const float u = ((float)px / (float)W) * 2.0f - 1.0f;
        const float x = row(0, u, 1.0f);
        if (!same(x, row(0, u, -1.0f))) return false;
        const float q = divide ? x / w : x;
        if (!(fabs((double)q) <= 1.0995e12)) return false;   // 2^40; also refuses NaN
        hi = fabs((double)q) > hi ? fabs((double)q) : hi;
        tab[(size_t)px] = q;
    }
    for (int py = 0; py < H; ++py) {
        const float v = ((float)py / (float)H) * 2.0f - 1.0f;
        const float y = row(1, 1.0f, v);
        if (!same(y, row(1, -1.0f, v))) return false;
        const float q = divide ? y / w : y;
        if (!(fabs((double)q) <= 1.0995e12)) return false;
        tab[(size_t)W + (size_t)py] = q;
    }
    z_out = divide ? z / w : z;
    // range of the first normalisation, dot = (x^2 + y^2) + z^2 >= z^2: inside [2^-80, 2^82], its root inside [2^-40, 2^41]
    const double az = fabs((double)z_out);
    return az >= 9.0949e-13 && az <= 1.0995e12;   // 2^-40 .. 2^40
}

// The second normalisation takes |invView3x3 * d| for a unit d: between the matrix' smallest and largest singular value.
// With F2 the squared Frobenius norm, sigma_max <= sqrt(F2) and sigma_min = |det| / (sigma_1 sigma_2) >= 2 |det| / F2.
// True when that keeps the squared length inside [2^-62, 2^42] with room for the rounding of the product.
bool view_matrix_in_range(const float *m) {
    double a[3][3], f2 = 0.0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) { a[r][c] = (double)m[c * 4 + r]; f2 += a[r][c] * a[r][c]; }
    if (!(f2 >= 9.0949e-13 && f2 <= 1.0995e12)) return false;      // 2^-40 .. 2^40, refuses NaN
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    return det * det >= 2.3842e-7 * f2 * f2 * f2;                   // sigma_min >= 2^-10 sqrt(F2)  <=  det^2 >= 2^-22 F2^3
}

// The view's table, from the cache or built and uploaded now (a synchronous 12 KB copy, once per projection and frame
// shape). nullptr: this projection has no table. Eight tables are kept; the least recently used one is replaced after a
// device synchronize (launches on any stream may still read it).
vrt_ctx::RayTable *ray_table(vrt_ctx *c, const float *inv_proj, int W, int H) {
    if (!c->ray_tables_on) return nullptr;
    vrt_ctx::RayTable *hit = nullptr, *lru = nullptr;
    for (auto &t : c->ray_tables) {
        if (t.width == W && t.height == H && std::memcmp(t.inv_proj, inv_proj, sizeof t.inv_proj) == 0) hit = &t;
        if (!lru || t.last_use < lru->last_use) lru = &t;
    }
    if (hit) {
        hit->last_use = ++c->ray_tick;
        return hit->ok ? hit : nullptr;
    }
    std::vector<float> tab;
    float z = 0.0f;
    const bool ok = build_ray_table(inv_proj, W, H, tab, z);
    vrt_ctx::RayTable *t;
    if (c->ray_tables.size() < 8) {
        c->ray_tables.emplace_back();
        t = &c->ray_tables.back();
    } else {
        t = lru;
        if (t->ok && hipDeviceSynchronize() != hipSuccess) return nullptr;
    }
    std::memcpy(t->inv_proj, inv_proj, sizeof t->inv_proj);
    t->width = W; t->height = H; t->z = z; t->ok = false;
    t->last_use = ++c->ray_tick;
    if (!ok) return nullptr;
    if (tab.size() > t->capacity) {
        float *fresh = nullptr;
        if (hipMalloc((void **)&fresh, tab.size() * sizeof(float)) != hipSuccess) return nullptr;
        if (t->d_tab) (void)hipFree(t->d_tab);
        t->d_tab = fresh;
        t->capacity = tab.size();
    }
    if (hipMemcpy(t->d_tab, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    t->ok = true;
    return t;
}

// Builds the kernel arguments for local rows [0, n_rows) and enqueues one launch.
int ensure_analysis(vrt_ctx *c);

// views == nullptr: one view, the context's camera (vrt_set_camera) rendering into d_rgba / d_id.
int enqueue(vrt_ctx *c, int width, int height, int row0, int n_rows, int tile_rows, int row_stride, int compact,
            int mode, void *d_rgba, void *d_id, hipStream_t s, const vrt_view *views = nullptr, int n_views = 1) {
    if (!c->have_scene) return fail(c, VRT_E_STATE, "vrt_dispatch: no octree uploaded (call vrt_upload_octree first)");
    if (c->batch.open) return fail(c, VRT_E_STATE, "vrt_dispatch: a patch batch is open (call vrt_patch_end first)");
    if (!views && !c->have_camera) return fail(c, VRT_E_STATE, "vrt_dispatch: no camera set (call vrt_set_camera first)");
    if (n_views < 1 || n_views > vrt::kMaxViews) return fail(c, VRT_E_INVALID, "vrt_dispatch_views: 1 to 4 views per launch");
    if (mode != VRT_MODE_PRIMARY && mode != VRT_MODE_PRIMARY_SHADOW && mode != VRT_MODE_FULL)
        return fail(c, VRT_E_INVALID, "unknown mode");
    if (n_rows <= 0) return VRT_OK;
    {
        const int ra = ensure_analysis(c);
        if (ra) return ra;
    }
    Variant v = kVariants[c->variant];
    if (v.trav >= 3 && !c->wide_ok) {  // wide layout not expressible for this scene: record-array kernels
        v.trav = 2; v.use_lds = false; v.tw = 8; v.block = 256; v.wpe = 1; v.lds_cap = 0;
    }
    if (v.trav == 2 && c->unit_internal) {  // precondition of vrt_kernels.hip.h not met: explicit-AABB kernels
        v.trav = 1; v.tw = 8; v.block = 256; v.wpe = 1;
        if (v.lds_cap > 2048) v.lds_cap = 2048;
    }
    if (mode == VRT_MODE_FULL) {
        // the full path tracer exists for the wide traversal (64- or 256-lane workgroups, five waves per SIMD: 96 VGPRs
        // and no extra spills measured 8-10 % faster than the unconstrained 105-VGPR build) and, as baselines, for
        // the other two in one shape each
        v.use_lds = false; v.tw = 8; v.lds_cap = 0; v.blocks_per_cu = 0;
        // the default takes the v4 traversal here too (one march loop, for rays that start in any medium: 96 registers
        // without spills; 9 % faster than v3, profiles/r02_f_full_shader_v4_ab.jsonl); variant 20 is v3, and so is the
        // two-kernel experiment of an A/B build
        if (v.trav == 4 && VRT_AB && c->full_split) v.trav = 3;
        if (v.trav >= 3) { v.block = (v.block == 64 || !VRT_AB) ? 64 : 256; v.wpe = 5; }
        else { v.block = 256; v.wpe = 1; }
    } else if (mode == VRT_MODE_PRIMARY_SHADOW && c->variant == 20 && v.trav == 3) {
        v.wpe = 7;  // round 1's default: the shadow march was 1.5 % faster seven waves deep, the primary one six deep
    }
    vrt::KArgs a;
    vrt::ViewSet vs;
    std::memset(&vs, 0, sizeof vs);
    a.n_views = n_views;
    int eyes[vrt::kMaxViews][3];
    for (int i = 0; i < n_views; ++i) {
        vrt::View &w = vs.v[i];
        std::memcpy(w.inv_proj, views ? views[i].inv_projection : c->inv_proj, sizeof w.inv_proj);
        std::memcpy(w.inv_view, views ? views[i].inv_view : c->inv_view, sizeof w.inv_view);
        std::memcpy(w.cam_pos, views ? views[i].camera_pos : c->cam_pos, sizeof w.cam_pos);
        w.out_rgba = (uint32_t *)(views ? views[i].d_rgba8 : d_rgba);
        w.out_id = (int2 *)(views ? views[i].d_id_dist : d_id);
        // the shader's lookup at the eye (comp:445-449), same arithmetic: floor(cameraPos * u_voxelScale)
        int eye[3];
        for (int k = 0; k < 3; ++k) {
            const float g = floorf(w.cam_pos[k] * c->params.voxel_scale);
            // float -> int as the device converts: NaN -> 0, out of range saturates (and is outside any world)
            eye[k] = g != g ? 0 : (g >= 2147483648.0f ? 2147483647 : (g < -2147483648.0f ? (-2147483647 - 1) : (int)g));
        }
        for (int k = 0; k < 3; ++k) eyes[i][k] = eye[k];
        vrt::eye_lookup(c->host_records, c->params.world_min, c->params.world_max, eye, w.eye0, w.eye1);
        vrt::FirstFind ff;
        w.first_valid = (c->wide_ok && vrt::first_find(c->wide, c->params.world_min, c->params.world_max, eye, vrt::v3::kAnchorShift, ff)) ? 1 : 0;
        if (w.first_valid) {
            w.first_w0 = ff.w0; w.first_w1 = ff.w1; w.first_node = ff.node; w.first_anode = ff.anode;
            w.first_s = ff.s; w.first_as = ff.as;
        }
        w.gen_x = w.gen_y = nullptr; w.gen_z = 0.0f; w.gen_fast = 0u;
        if (view_matrix_in_range(w.inv_view)) {
            if (const vrt_ctx::RayTable *t = ray_table(c, w.inv_proj, width, height)) {
                w.gen_x = t->d_tab; w.gen_y = t->d_tab + width; w.gen_z = t->z; w.gen_fast = 1u;
            }
        }
    }
    if (v.trav == 4 && mode != VRT_MODE_FULL) {
        // the v4 primary kernels hold the march loop for rays that start in refraction byte 85 (1.0) only: an eye inside a
        // medium (comp:445-449: refraction byte 1..254 of the voxel that holds it) takes the v3 kernels
        bool eye_in_medium = false;
        for (int i = 0; i < n_views; ++i) {
            const uint32_t b = vs.v[i].eye1 & 0xffu;
            eye_in_medium = eye_in_medium || (b >= 1u && b <= 254u && b != 85u);
        }
        if (eye_in_medium) { v.trav = 3; v.wpe = 6; }
    }
    a.voxel_scale = c->params.voxel_scale;
    for (int i = 0; i < 3; ++i) {
        a.wmin[i] = c->params.world_min[i];
        a.wmax[i] = c->params.world_max[i];
        a.light_dir[i] = c->params.light_dir[i];
        a.highlighted[i] = c->params.highlighted[i];
        // comp:335-345 on the launch's one light direction
        const float d = a.light_dir[i];
        a.light_inv[i] = (fabsf(d) < 1e-8f) ? 1e20f : 1.0f / d;
        a.light_push[i] = (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) * 0.001f;
        a.light_dpos[i] = d > 0.0f ? 1 : 0;
        a.light_dposf[i] = d > 0.0f ? 1.0f : 0.0f;
    }
    for (int i = 0; i < 4; ++i) a.global_light[i] = c->params.global_light[i];
    // |globalLight|, |lightDir| <= 2^30: direct (light * n.l) * colour * throughput (starts as the light) stays below 2^90 < 2^97,
    // where x / PI needs no range scaling
    a.shade_fast = 1;
    for (int i = 0; i < 3; ++i)
        if (!(fabsf(a.global_light[i]) <= 1073741824.0f) || !(fabsf(a.light_dir[i]) <= 1073741824.0f)) a.shade_fast = 0;
    a.tex_dim = (int)c->info.tex_dim;
    a.width = width;
    a.height = height;
    a.row0 = row0;
    a.n_rows = n_rows;
    a.tile_rows = tile_rows;
    a.row_stride = row_stride;
    a.compact = compact;
    a.nodes = c->d_nodes;
    a.n_records = c->info.n_records;
    a.lds_records = v.use_lds ? (c->info.n_records < v.lds_cap ? c->info.n_records : v.lds_cap) : 0u;
    a.cells = c->d_cells;
    a.cells4 = c->d_cells ? c->d_cells + c->cells_capacity : nullptr;
    a.n_roots = c->wide_ok ? (uint32_t)c->wide.roots.size() : 0u;
    for (int k = 0; k < 3; ++k) a.root0_min[k] = a.n_roots ? c->wide.roots[0].origin[k] : 0;
    a.root_table = c->d_roots;
    a.root0_node = a.n_roots ? c->wide.roots[0].node : 0u;
    a.root0_shift = a.n_roots ? c->wide.roots[0].shift : 0;
    // nothing outside wide root 0? (the shipped maps: the octree root's only child is the octant [0, 1024)^3) -- then rays
    // that leave it are done (find() in vrt_kernels_v4.hip.h), and the same argument one level down, as often as it holds,
    // lets a deeper node stand in for it: a shorter descent whenever a lookup restarts there, leaving rays done sooner.
    // dragon.vox: [0, 1024)^3 holds everything in its cell [0, 256)^3, whose 64-unit cells the model spreads over: root 0
    // becomes [0, 256)^3 for eyes inside it. Not below the anchor level (a node of side 2^kAnchorShift).
    a.root0_only = (a.n_roots == 1u && c->root0_only_on && vrt::content_only_in_root0(c->host_records, c->wide)) ? 1 : 0;
    if (a.root0_only && c->tight_root_on)
        vrt::tighten_root0(c->wide, eyes, n_views, vrt::v3::kAnchorShift, a.root0_node, a.root0_shift, a.root0_min);
    a.group_order = nullptr;
    a.tile_cost = nullptr;
    a.defer_rec = nullptr;
    a.defer_count = nullptr;
    a.defer_cap = 0;

    const int th = 64 / v.tw;
    const long tiles = (long)((width + v.tw - 1) / v.tw) * (long)((n_rows + th - 1) / th);
    {   // index arithmetic of the prologue without integer divisions where the shapes allow it
        const unsigned long tiles_x = (unsigned long)((width + v.tw - 1) / v.tw);
        // q = (n * M) >> 32 with M = floor(2^32 / d) + 1 equals n / d while n * d < 2^32 (the error term n * (M * d - 2^32) stays below 2^32)
        a.tiles_x_magic = (tiles_x > 1 && (unsigned long)(tiles + 4) * tiles_x < (1ul << 32)) ? (uint32_t)((1ul << 32) / tiles_x + 1) : 0u;
        a.row_mode = tile_rows >= n_rows ? 1 : ((tile_rows == 8 && th == 8) ? 2 : 0);
    }
    const int waves = v.block / 64;
    long grid = (tiles + waves - 1) / waves;
    if (v.blocks_per_cu > 0) {
        long cap = (long)c->n_cus * v.blocks_per_cu;
        if (grid > cap) grid = cap;
    }
    if (grid < 1) grid = 1;
    const size_t lds_bytes = (size_t)a.lds_records * sizeof(uint2);
    // feedback scheduling: wide-traversal kernels, one view, launches large enough to have a tail worth shaping
    SchedState *st = nullptr;
    bool measure = false;
    const bool sched_kernel = v.trav >= 3 && !v.use_lds && v.tw == 8 && v.blocks_per_cu == 0 && n_views == 1 &&
                              (v.trav == 4 || (v.block == 64 && (v.wpe == 5 || v.wpe == 6 || v.wpe == 7)) ||
                               (v.block == 256 && (v.wpe == 5 || v.wpe == 6)));
    const long groups = (tiles + vrt::kGroupTiles - 1) / vrt::kGroupTiles;
    if (sched_kernel && c->dbg_sched) {
        a.group_order = c->dbg_group_order;
        a.tile_cost = c->dbg_tile_cost;
    } else if (sched_kernel && c->sched_period > 0 && groups >= kSchedMinGroups && groups <= kSchedMaxGroups) {
        st = sched_state(c, s, width, n_rows, row0, row_stride, tile_rows, mode, (uint32_t)tiles, (uint32_t)groups);
        if (st) {
            // eye = invView's translation column, viewing direction = minus its third column (column-major)
            const float *iv = vs.v[0].inv_view;
            const float now[6] = {iv[12], iv[13], iv[14], -iv[8], -iv[9], -iv[10]};
            // the first launch of a shape is never the one measured: it may be the process's first launch of the kernel
            // (code object load, cold instruction cache and TLB), and its tile times would shape the next period's order
            measure = measuring_launch(st->launches, c->sched_period) || (st->valid && camera_jumped(st->cam, now));
            if (measure) std::memcpy(st->cam, now, sizeof now);
            a.group_order = st->valid ? st->d_order : nullptr;
            a.tile_cost = measure ? st->d_cost : nullptr;
        }
    }
    if (a.group_order) grid = groups * (vrt::kGroupTiles / waves);  // whole groups: the last one may hold tiles past the end
    const bool prof = c->profiling && (c->prof_seen++ % c->prof_stride) == 0 && (c->prof_count + 1) * 2 <= c->prof_events.size();
    const hipEvent_t ev0 = prof ? c->prof_events[2 * c->prof_count] : nullptr;
    const hipEvent_t ev1 = prof ? c->prof_events[2 * c->prof_count + 1] : nullptr;
    hipError_t e;
    // The full path tracer as two kernels (vrt_bounce.hip.h): the default traversal, one view, a scene with a wide form.
    const bool split = VRT_AB && mode == VRT_MODE_FULL && c->full_split && c->wide_ok && v.trav == 3 && v.block == 64 && n_views == 1 && c->variant == 0;
    if (split) {
        const size_t cap = (size_t)((tiles + vrt::kDeferQueues - 1) / vrt::kDeferQueues) * 64;   // every pixel of a queue's tiles may defer
        vrt_ctx::DeferQueues *dq = nullptr;
        for (auto &d : c->defer)
            if (d.stream == s) dq = &d;
        if (!dq) {
            if (c->defer.size() < 8) {
                c->defer.emplace_back();
                dq = &c->defer.back();
            } else {   // recycle the least recently used set: its launches may still be in flight on its stream
                for (auto &d : c->defer)
                    if (!dq || d.last_use < dq->last_use) dq = &d;
                VRT_HIP(c, hipStreamSynchronize(dq->stream));
            }
            dq->stream = s;
        }
        if (cap > dq->cap) {
            VRT_HIP(c, hipStreamSynchronize(s));   // launches in flight on this stream still use the old queues
            float *fresh = nullptr;
            VRT_HIP(c, hipMalloc((void **)&fresh, cap * vrt::kDeferQueues * vrt::kDeferPlanes * sizeof(float)));
            if (dq->rec) (void)hipFree(dq->rec);
            dq->rec = fresh;
            dq->cap = cap;
        }
        if (!dq->count) VRT_HIP(c, hipMalloc((void **)&dq->count, 2 * vrt::kDeferQueues * vrt::kDeferStride * sizeof(uint32_t)));
        dq->last_use = ++c->defer_tick;
        VRT_HIP(c, hipMemsetAsync(dq->count, 0, 2 * vrt::kDeferQueues * vrt::kDeferStride * sizeof(uint32_t), s));
        a.defer_rec = dq->rec;
        a.defer_count = dq->count;
        a.defer_cap = (uint32_t)dq->cap;
    }
#if VRT_AB
    if (split) {
        e = launch_sched<3, vrt::v3::Trav, 8, 64, 5>(a, vs, (int)grid, 0, s, ev0, nullptr);
        if (e == hipSuccess) {
            vrt::bounce::Args b;
            b.out_rgba = vs.v[0].out_rgba;
            b.refill_below = c->bounce_refill_below;
            if (b.out_rgba) {   // without a colour image there is nothing for the bounce rays to finish
                const int waves = c->n_cus * 4 * c->bounce_waves_per_simd;   // the chip filled once (six waves per SIMD: the kernel's register budget)
                if (ev1) hipExtLaunchKernelGGL(vrt::bounce::bounce_kernel, dim3(waves), dim3(64), 0, s, nullptr, ev1, 0, a, b);
                else hipLaunchKernelGGL(vrt::bounce::bounce_kernel, dim3(waves), dim3(64), 0, s, a, b);
                e = hipGetLastError();
            } else if (ev1) {
                e = hipEventRecord(ev1, s);
            }
        }
    } else
#endif
    if (mode == VRT_MODE_FULL) {
        if (v.trav == 4) e = launch_sched<2, vrt::v4::TravAny, 8, 64, 5>(a, vs, (int)grid, 0, s, ev0, ev1);
        else if (v.trav == 3 && v.block == 64) e = launch_sched<2, vrt::v3::Trav, 8, 64, 5>(a, vs, (int)grid, 0, s, ev0, ev1);
#if VRT_AB
        else if (v.trav == 3) e = launch_sched<2, vrt::v3::Trav, 8, 256, 5>(a, vs, (int)grid, 0, s, ev0, ev1);
#endif
        else if (v.trav == 2) e = launch_one<2, vrt::v2::Trav<false>, 8, 256, 1>(a, vs, (int)grid, 0, s);
        else e = launch_one<2, vrt::v1::Trav<false>, 8, 256, 1>(a, vs, (int)grid, 0, s);
    } else {
        e = (mode == VRT_MODE_PRIMARY) ? launch_mode<0>(v, a, vs, (int)grid, lds_bytes, s, ev0, ev1)
                                       : launch_mode<1>(v, a, vs, (int)grid, lds_bytes, s, ev0, ev1);
    }
    if (e != hipSuccess) return fail(c, VRT_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    if (st) {
        ++st->launches;
        if (measure) {
            const int rr = launch_order_kernel(c, st, s);
            if (rr) return rr;
        }
    }
    if (prof) ++c->prof_count;
    c->info.lds_records = a.lds_records;
    return VRT_OK;
}

int check_frame(vrt_ctx *c, int width, int height) {
    if (!c) return VRT_E_INVALID;
    if (width < 1 || height < 1 || (long)width * (long)height > (1L << 30))
        return fail(c, VRT_E_INVALID, "width/height out of range");
    return VRT_OK;
}

}  // namespace

extern "C" {

const char *vrt_version(void) { return "vrt-hip 0.1 (gfx950)"; }

void vrt_default_params(vrt_params *p) {
    if (!p) return;
    p->voxel_scale = 1.0f;  // src/main.cpp:638
    for (int i = 0; i < 3; ++i) {
        p->world_min[i] = -1023;  // src/main.cpp:478-480
        p->world_max[i] = 1024;
        p->highlighted[i] = -1;   // src/main.cpp:816
    }
    for (int i = 0; i < 4; ++i) p->global_light[i] = 1.0f;  // src/main.cpp:482
    // glm::normalize(vec3(0.3481553, 0.870388, 0.3481553)), src/main.cpp:483
    const float l[3] = {0.3481553f, 0.870388f, 0.3481553f};
    const float t0 = l[0] * l[0], t1 = l[1] * l[1], t2 = l[2] * l[2];
    const float inv = 1.0f / sqrtf(t0 + t1 + t2);
    for (int i = 0; i < 3; ++i) p->light_dir[i] = l[i] * inv;
}

int vrt_create(int device_id, vrt_ctx **out) {
    if (!out) return VRT_E_INVALID;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("vrt_create: no HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                         "); this library has no CPU path";
        return VRT_E_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "vrt_create: device_id out of range";
        return VRT_E_INVALID;
    }
    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    if (!c) return VRT_E_INVALID;
    c->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_error = std::string("vrt_create: ") + hipGetErrorString(e);
        delete c;
        return VRT_E_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
    vrt_default_params(&c->params);
    {   // the kernels re-read part of their arguments from the kernarg segment (late_args / late_view): check the layout
        // they assume on this device before anything depends on it
        vrt::KArgs a;
        vrt::ViewSet vs;
        std::memset(&a, 0, sizeof a);
        std::memset(&vs, 0, sizeof vs);
        a.n_views = vrt::kMaxViews; a.width = 0x1234; a.height = 0x2345; a.tex_dim = 77; a.compact = 1; a.voxel_scale = 0.75f;
        a.light_dir[2] = 0.5f; a.highlighted[1] = -9;
        for (int i = 0; i < vrt::kMaxViews; ++i) {
            vs.v[i].out_rgba = (uint32_t *)(uintptr_t)(0x1000u + 16u * (unsigned)i);
            vs.v[i].out_id = (int2 *)(uintptr_t)(0x2000u + 16u * (unsigned)i);
            vs.v[i].cam_pos[1] = 3.0f + (float)i;
        }
        uint32_t *d_bad = nullptr, bad = 1;
        if ((e = hipMalloc((void **)&d_bad, sizeof bad)) == hipSuccess && (e = hipMemsetAsync(d_bad, 0, sizeof bad, c->stream)) == hipSuccess) {
            hipLaunchKernelGGL(vrt::kernarg_probe_kernel, dim3(1, vrt::kMaxViews), dim3(64), 0, c->stream, a, vs, d_bad);
            if ((e = hipGetLastError()) == hipSuccess && (e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream)) == hipSuccess)
                e = hipStreamSynchronize(c->stream);
        }
        (void)hipFree(d_bad);
        if (e != hipSuccess || bad != 0) {
            g_create_error = e != hipSuccess ? std::string("vrt_create: kernarg probe: ") + hipGetErrorString(e)
                                             : "vrt_create: the kernarg segment is not laid out as late_args()/late_view() assume";
            (void)hipStreamDestroy(c->stream);
            delete c;
            return e != hipSuccess ? VRT_E_NO_DEVICE : VRT_E_HIP;
        }
    }
    *out = c;
    return VRT_OK;
}

void vrt_destroy(vrt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_nodes) (void)hipFree(c->d_nodes);
    if (c->d_cells) (void)hipFree(c->d_cells);
    if (c->d_roots) (void)hipFree(c->d_roots);
    if (c->d_rgba) (void)hipFree(c->d_rgba);
    if (c->d_id) (void)hipFree(c->d_id);
    if (c->d_shown) (void)hipFree(c->d_shown);
    for (auto &d : c->defer) {
        (void)hipFree(d.rec);
        (void)hipFree(d.count);
    }
    if (!c->sched.empty()) (void)hipDeviceSynchronize();  // their launches may be on the caller's streams
    for (SchedState &st : c->sched) {
        (void)hipFree(st.d_cost);
        (void)hipFree(st.d_order);
    }
    for (auto &ln : c->lane) {
        if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
        (void)hipFree(ln.d_rgba);
        (void)hipFree(ln.d_id);
        if (ln.done) (void)hipEventDestroy(ln.done);
    }
    for (auto &e : c->prof_events) (void)hipEventDestroy(e);
    if (!c->ray_tables.empty()) (void)hipDeviceSynchronize();
    for (auto &t : c->ray_tables) (void)hipFree(t.d_tab);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *vrt_last_error(const vrt_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int vrt_set_params(vrt_ctx *c, const vrt_params *p) {
    if (!c || !p) return c ? fail(c, VRT_E_INVALID, "vrt_set_params: null params") : VRT_E_INVALID;
    for (int i = 0; i < 3; ++i)
        if (p->world_max[i] < p->world_min[i]) return fail(c, VRT_E_INVALID, "vrt_set_params: world_max < world_min");
    c->params = *p;
    c->analysis_valid = false;
    return VRT_OK;
}

namespace {
// src/main.cpp:266-268: tex_dim = (size_t)ceil(cbrt((double)total_texels)), at least 1
uint32_t dim_of_texels(size_t texels) {
    const size_t d = (size_t)ceil(cbrt((double)texels));
    return (uint32_t)(d == 0 ? 1 : d);
}
}  // namespace

int vrt_upload_octree(vrt_ctx *c, const uint8_t *texels, size_t used_bytes, uint32_t tex_dim) {
    if (!c) return VRT_E_INVALID;
    if (used_bytes % 4 != 0) return fail(c, VRT_E_INVALID, "vrt_upload_octree: used_bytes must be a multiple of 4");
    if (used_bytes / 4 > (1u << 23)) return fail(c, VRT_E_MALFORMED, "vrt_upload_octree: more than 2^23 texels cannot be addressed by 23-bit node pointers");
    if (tex_dim == 0) tex_dim = 1;
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return fail(c, VRT_E_MALFORMED, "vrt_upload_octree: " + err);
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = lay.records.size() * sizeof(vrt::Record);
    if (bytes > c->nodes_capacity) {
        VRT_HIP(c, hipDeviceSynchronize());   // dispatches still reading the old array, on whatever stream
        uint2 *fresh = nullptr;
        VRT_HIP(c, hipMalloc((void **)&fresh, bytes));   // before the old array goes: a failure leaves the context as it was
        if (c->d_nodes) (void)hipFree(c->d_nodes);
        c->d_nodes = fresh;
        c->nodes_capacity = bytes;
    }
    // after every dispatch still reading the old tree, on whatever stream the caller enqueued it (uploads are rare:
    // a device-wide wait is cheaper than a contract about foreign streams); synchronous so `lay` may die
    VRT_HIP(c, hipDeviceSynchronize());
    VRT_HIP(c, hipMemcpyAsync(c->d_nodes, lay.records.data(), bytes, hipMemcpyHostToDevice, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    c->info.tex_dim = tex_dim;
    c->info.n_texels = (uint32_t)(used_bytes / 4);
    c->info.n_records = (uint32_t)lay.records.size();
    c->info.n_internal = lay.n_internal;
    c->info.n_leaves = lay.n_leaves;
    c->info.max_depth = lay.max_depth;
    c->info.lds_records = 0;
    c->host_records.swap(lay.records);
    c->uploaded_records = c->host_records.size();
    c->stream_texels = used_bytes / 4;
    c->dim_from_texels = tex_dim == dim_of_texels(c->stream_texels);
    c->analysis_valid = false;
    c->have_scene = true;
    return VRT_OK;
}

// Extension beyond the reference boundary: take the device record array (vrt_layout.h) directly, as
// libvrt_host.so emits it from the pointer octree (vrth_world_records). Skips the texel stream, and with
// it the stream's 23-bit pointer limit and the flatten + re-parse on every edit.
int vrt_upload_records(vrt_ctx *c, const uint32_t *records, size_t n_records, uint32_t tex_dim) {
    if (!c) return VRT_E_INVALID;
    if (!records || n_records == 0 || n_records > (1ull << 31)) return fail(c, VRT_E_INVALID, "vrt_upload_records: bad record array");
    if (tex_dim == 0) tex_dim = 1;
    // structural check: every child index lies after its parent (level order) and inside the array, so a
    // descent always terminates; depth is bounded by the same 16-iteration rule as the texel path
    std::vector<vrt::Record> recs(n_records);
    std::memcpy(recs.data(), records, n_records * sizeof(vrt::Record));
    std::vector<uint8_t> kind(n_records, 0);  // 1 internal, 2 leaf
    std::vector<uint8_t> depth(n_records, 0);
    kind[0] = 1;
    uint32_t n_internal = 0, n_leaves = 0, max_depth = 0;
    for (size_t i = 0; i < n_records; ++i) {
        if (kind[i] != 1) { if (kind[i] == 2) ++n_leaves; continue; }
        ++n_internal;
        uint32_t mask = recs[i].w0 & 0xffu;
        const uint32_t leaf_mask = (recs[i].w0 >> 8) & 0xffu, base = recs[i].w1;
        if (depth[i] >= 15) { recs[i].w0 = 0; mask = 0; }
        const uint32_t n_child = (uint32_t)__builtin_popcount(mask);
        if (n_child == 0) continue;
        if (base <= i || (size_t)base + n_child > n_records) return fail(c, VRT_E_MALFORMED, "vrt_upload_records: child index out of order or range");
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t idx = (size_t)base + rank++;
            if (kind[idx] != 0) return fail(c, VRT_E_MALFORMED, "vrt_upload_records: a record has two parents");
            kind[idx] = ((leaf_mask >> ci) & 1u) ? 2 : 1;
            depth[idx] = (uint8_t)(depth[i] + 1);
            if (depth[idx] > max_depth) max_depth = depth[idx];
        }
    }
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t bytes = n_records * sizeof(vrt::Record);
    if (bytes > c->nodes_capacity) {
        VRT_HIP(c, hipDeviceSynchronize());   // dispatches still reading the old array, on whatever stream
        uint2 *fresh = nullptr;
        VRT_HIP(c, hipMalloc((void **)&fresh, bytes));   // before the old array goes: a failure leaves the context as it was
        if (c->d_nodes) (void)hipFree(c->d_nodes);
        c->d_nodes = fresh;
        c->nodes_capacity = bytes;
    }
    VRT_HIP(c, hipDeviceSynchronize());  // see vrt_upload_octree
    VRT_HIP(c, hipMemcpyAsync(c->d_nodes, recs.data(), bytes, hipMemcpyHostToDevice, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    c->info.tex_dim = tex_dim;
    c->info.n_texels = 0;
    c->info.n_records = (uint32_t)n_records;
    c->info.n_internal = n_internal;
    c->info.n_leaves = n_leaves;
    c->info.max_depth = max_depth;
    c->info.lds_records = 0;
    c->host_records.swap(recs);
    c->uploaded_records = c->host_records.size();
    c->stream_texels = vrt::stream_texels(c->host_records.data(), c->host_records.size(), 0);
    c->dim_from_texels = tex_dim == dim_of_texels(c->stream_texels);
    c->analysis_valid = false;
    c->have_scene = true;
    return VRT_OK;
}

namespace {
// (re)derives what depends on the world bounds: whether the wide layout can be used, and the layout itself on the
// device. Called lazily by the dispatcher and by the patch entry points.
// the wide roots' table on the device; blocking (callers have synchronised the device or run before any dispatch)
// The wide cells live on the device twice: as vrt_layout.h lays them out (v3 kernels, tests) and in the form the v4
// kernels read (vrt::to_cell4), behind it in the same allocation. cells_capacity counts CELLS of one form.
int reserve_cells(vrt_ctx *c, size_t n_cells) {
    if (n_cells <= c->cells_capacity) return VRT_OK;
    uint2 *fresh = nullptr;
    VRT_HIP(c, hipMalloc((void **)&fresh, 2 * n_cells * sizeof(uint2)));   // before the old one goes: a failure leaves the context usable
    if (c->d_cells) (void)hipFree(c->d_cells);
    c->d_cells = fresh;
    c->cells_capacity = n_cells;
    return VRT_OK;
}

// cells [from, from + n) of c->wide to the device in both forms; blocking (callers have synchronised the device)
int upload_cells(vrt_ctx *c, size_t from, size_t n) {
    if (n == 0) return VRT_OK;
    if (from + n > c->wide.cells.size() || from + n > c->cells_capacity) return fail(c, VRT_E_STATE, "upload_cells: range outside the wide layout");
    VRT_HIP(c, hipMemcpy(c->d_cells + from, c->wide.cells.data() + from, n * sizeof(vrt::WideCell), hipMemcpyHostToDevice));
    std::vector<vrt::WideCell> c4(n);
    for (size_t i = 0; i < n; ++i) c4[i] = vrt::to_cell4(c->wide.cells[from + i]);
    VRT_HIP(c, hipMemcpy(c->d_cells + c->cells_capacity + from, c4.data(), n * sizeof(vrt::WideCell), hipMemcpyHostToDevice));
    return VRT_OK;
}

int upload_roots(vrt_ctx *c) {
    uint32_t t[16];
    for (int i = 0; i < 8; ++i) {
        const bool on = c->wide_ok && (size_t)i < c->wide.roots.size();
        t[i] = on ? c->wide.roots[(size_t)i].record : 0xffffffffu;
        t[8 + i] = on ? c->wide.roots[(size_t)i].node : 0u;
    }
    if (!c->d_roots) VRT_HIP(c, hipMalloc((void **)&c->d_roots, sizeof t));
    VRT_HIP(c, hipMemcpy(c->d_roots, t, sizeof t, hipMemcpyHostToDevice));
    return VRT_OK;
}

int ensure_analysis(vrt_ctx *c) {
    if (c->analysis_valid) return VRT_OK;
    c->unit_internal = vrt::has_unit_internal_node(c->host_records, c->params.world_min, c->params.world_max);
    std::string why;
    c->wide_ok = !c->unit_internal &&
                 vrt::build_wide(c->host_records, c->params.world_min, c->params.world_max, c->wide, why);
    if (c->wide_ok) {
        const size_t n_cells = c->wide.cells.empty() ? 64 : c->wide.cells.size();
        // rare (scene or bounds changed): blocking copies keep it ordered against any caller stream
        VRT_HIP(c, hipDeviceSynchronize());
        if (n_cells > c->cells_capacity) {
            const int rr = reserve_cells(c, n_cells + n_cells / 2);  // room for patches
            if (rr) { c->have_scene = false; return rr; }
        }
        const int rr = upload_cells(c, 0, c->wide.cells.size());
        if (rr) return rr;
    }
    VRT_HIP(c, hipDeviceSynchronize());
    {
        const int rr = upload_roots(c);
        if (rr) return rr;
    }
    c->analysis_valid = true;
    return VRT_OK;
}
}  // namespace

int vrt_patch_plan(vrt_ctx *c, int x, int y, int z, int max_depth, vrt_patch *out) {
    if (!c || !out) return c ? fail(c, VRT_E_INVALID, "vrt_patch_plan: null output") : VRT_E_INVALID;
    if (!c->have_scene) return fail(c, VRT_E_STATE, "vrt_patch_plan: no octree uploaded");
    VRT_HIP(c, hipSetDevice(c->device));
    int r = ensure_analysis(c);
    if (r) return r;
    // replaced sub-trees stay allocated: once they outweigh the tree, reclaim them (the live tree re-laid, wide layout
    // rebuilt) instead of asking the caller for a full upload
    if (!c->batch.open && c->host_records.size() > 2 * c->uploaded_records + (1u << 12)) {
        r = vrt_compact(c);
        if (!r) r = ensure_analysis(c);
        if (r) return r;
    }
    vrt::PatchSite site;
    const int p[3] = {x, y, z};
    if (!vrt::plan_patch(c->host_records, c->wide, c->wide_ok && !c->batch.wide_invalid, c->params.world_min, c->params.world_max, p,
                         max_depth > 15 ? 15 : max_depth, site))
        return fail(c, VRT_E_STATE, "vrt_patch_plan: no patchable ancestor (full upload needed)");
    out->depth = site.depth;
    std::memcpy(out->path, site.path, sizeof out->path);
    return VRT_OK;
}

namespace {
struct DropSceneOnFailure {   // host structures ahead of the device copies: a failure must not leave a context that would
    vrt_ctx *c;               // dispatch over half-updated or freed arrays, so it drops the scene (the caller uploads again)
    bool armed = true;
    ~DropSceneOnFailure() { if (armed) { c->have_scene = false; c->analysis_valid = false; c->batch = vrt_ctx::PatchBatch(); } }
};

// the host half of one patch: validates the plan against the structures as they are NOW, applies it, notes what changed
int patch_host(vrt_ctx *c, const vrt_patch &patch, const uint32_t *subtree_records, size_t n_records) {
    vrt_ctx::PatchBatch &bt = c->batch;
    vrt::PatchSite site;
    int lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = c->params.world_min[k]; hi[k] = c->params.world_max[k]; }
    for (int d = 0; d < patch.depth; ++d) {   // find A again from the path (the plan carries no pointers into this context)
        const uint32_t ci = patch.path[d];
        if (ci > 7) return fail(c, VRT_E_INVALID, "vrt_patch_apply: bad path");
        for (int k = 0; k < 3; ++k) {
            const int mid = lo[k] + ((hi[k] - lo[k]) >> 1);
            if ((ci >> (2 - k)) & 1u) lo[k] = mid; else hi[k] = mid;
        }
    }
    // a patch that voided the wide layout earlier in the batch leaves the rest to the record array alone
    const bool wide_now = c->wide_ok && !bt.wide_invalid;
    if (!vrt::plan_patch(c->host_records, c->wide, wide_now, c->params.world_min, c->params.world_max, lo, patch.depth, site) ||
        site.depth != patch.depth || std::memcmp(site.path, patch.path, (size_t)patch.depth) != 0)
        return fail(c, VRT_E_STATE, "vrt_patch_apply: the path does not name a patchable node of the uploaded tree");
    vrt::PatchRanges rg;
    std::string why;
    if (!vrt::apply_patch(c->host_records, c->wide, wide_now, site, reinterpret_cast<const vrt::Record *>(subtree_records), n_records, rg, why))
        return fail(c, VRT_E_MALFORMED, "vrt_patch_apply: " + why);   // apply_patch modifies nothing when it refuses
    bt.dirty = true;
    bt.rewritten_records.push_back(site.record);
    bt.texel_delta += rg.texel_delta;
    if (wide_now) {
        bt.wide_invalid = bt.wide_invalid || rg.wide_invalid;
        bt.roots_changed = bt.roots_changed || site.root_index >= 0;
        if (rg.cell_repointed) bt.repointed_cells.push_back((size_t)site.parent_node * 64 + site.parent_cell);
    }
    return VRT_OK;
}

// the device half: everything the batch appended or rewrote, after one wait for the dispatches in flight
int patch_device(vrt_ctx *c) {
    vrt_ctx::PatchBatch bt;
    std::swap(bt, c->batch);
    if (!bt.dirty) return VRT_OK;
    DropSceneOnFailure guard{c};
    VRT_HIP(c, hipDeviceSynchronize());
    const size_t rec_bytes = c->host_records.size() * sizeof(vrt::Record);
    if (rec_bytes > c->nodes_capacity) {
        uint2 *fresh = nullptr;
        VRT_HIP(c, hipMalloc((void **)&fresh, rec_bytes * 2));   // before the old array goes
        if (c->d_nodes) (void)hipFree(c->d_nodes);
        c->d_nodes = fresh;
        c->nodes_capacity = rec_bytes * 2;
        VRT_HIP(c, hipMemcpy(c->d_nodes, c->host_records.data(), rec_bytes, hipMemcpyHostToDevice));
    } else {
        if (c->host_records.size() > bt.records_before)
            VRT_HIP(c, hipMemcpy(c->d_nodes + bt.records_before, c->host_records.data() + bt.records_before,
                                 (c->host_records.size() - bt.records_before) * sizeof(vrt::Record), hipMemcpyHostToDevice));
        for (uint32_t rec : bt.rewritten_records)
            if (rec < bt.records_before)
                VRT_HIP(c, hipMemcpy(c->d_nodes + rec, c->host_records.data() + rec, sizeof(vrt::Record), hipMemcpyHostToDevice));
    }
    c->info.n_records = (uint32_t)c->host_records.size();
    c->stream_texels = (size_t)((long)c->stream_texels + bt.texel_delta);
    c->info.n_texels = (uint32_t)c->stream_texels;
    if (c->dim_from_texels) c->info.tex_dim = dim_of_texels(c->stream_texels);  // what updateGPUTexture would pass now
    if (c->wide_ok) {
        if (bt.wide_invalid) {
            c->analysis_valid = false;  // the next dispatch rebuilds the wide layout from the patched records
        } else {
            int rr = VRT_OK;
            if (c->wide.cells.size() > c->cells_capacity) {
                rr = reserve_cells(c, c->wide.cells.size() * 2);
                if (!rr) rr = upload_cells(c, 0, c->wide.cells.size());
            } else {
                if (c->wide.cells.size() > bt.cells_before) rr = upload_cells(c, bt.cells_before, c->wide.cells.size() - bt.cells_before);
                if (!rr && bt.roots_changed) rr = upload_roots(c);
                for (size_t at : bt.repointed_cells)
                    if (!rr && at < bt.cells_before) rr = upload_cells(c, at, 1);
            }
            if (rr) return rr;   // the guard drops the scene
        }
    }
    guard.armed = false;
    return VRT_OK;
}
}  // namespace

int vrt_patch_begin(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    if (!c->have_scene) return fail(c, VRT_E_STATE, "vrt_patch_begin: no octree uploaded");
    if (c->batch.open) return fail(c, VRT_E_STATE, "vrt_patch_begin: a batch is already open");
    VRT_HIP(c, hipSetDevice(c->device));
    const int r = ensure_analysis(c);
    if (r) return r;
    c->batch = vrt_ctx::PatchBatch();
    c->batch.open = true;
    c->batch.records_before = c->host_records.size();
    c->batch.cells_before = c->wide.cells.size();
    return VRT_OK;
}

int vrt_patch_end(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    if (!c->batch.open) return fail(c, VRT_E_STATE, "vrt_patch_end: no batch open");
    VRT_HIP(c, hipSetDevice(c->device));
    return patch_device(c);
}

int vrt_patch_apply(vrt_ctx *c, const vrt_patch *patch, const uint32_t *subtree_records, size_t n_records) {
    if (!c || !patch || !subtree_records) return c ? fail(c, VRT_E_INVALID, "vrt_patch_apply: null argument") : VRT_E_INVALID;
    if (!c->have_scene) return fail(c, VRT_E_STATE, "vrt_patch_apply: no octree uploaded");
    if (patch->depth < 1 || patch->depth > 15) return fail(c, VRT_E_INVALID, "vrt_patch_apply: depth out of range");
    const bool single = !c->batch.open;
    if (single) {
        const int r = vrt_patch_begin(c);
        if (r) return r;
    }
    DropSceneOnFailure guard{c};
    guard.armed = c->batch.dirty;   // a refused patch changes nothing: only a batch that already holds changes is lost with it
    int r = patch_host(c, *patch, subtree_records, n_records);
    if (r == VRT_OK && single) { guard.armed = false; return patch_device(c); }
    if (r != VRT_OK && !c->batch.dirty) { guard.armed = false; if (single) c->batch = vrt_ctx::PatchBatch(); }
    if (r == VRT_OK) guard.armed = false;
    return r;
}

// What patches leave behind goes: the live tree re-laid on the host, the device arrays replaced, the wide layout rebuilt
// at the next dispatch. No texel stream is involved; u_texDim and the stream's texel count are those the patches kept.
int vrt_compact(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    if (!c->have_scene) return fail(c, VRT_E_STATE, "vrt_compact: no octree uploaded");
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipDeviceSynchronize());   // dispatches in flight read the old arrays
    vrt::compact_records(c->host_records);
    struct Guard { vrt_ctx *c; bool armed = true; ~Guard() { if (armed) { c->have_scene = false; c->analysis_valid = false; } } } guard{c};
    const size_t bytes = c->host_records.size() * sizeof(vrt::Record);
    if (bytes > c->nodes_capacity) {   // cannot grow, but a context whose array was never sized stays correct
        uint2 *fresh = nullptr;
        VRT_HIP(c, hipMalloc((void **)&fresh, bytes));
        if (c->d_nodes) (void)hipFree(c->d_nodes);
        c->d_nodes = fresh;
        c->nodes_capacity = bytes;
    }
    VRT_HIP(c, hipMemcpy(c->d_nodes, c->host_records.data(), bytes, hipMemcpyHostToDevice));
    c->info.n_records = (uint32_t)c->host_records.size();
    c->uploaded_records = c->host_records.size();
    c->analysis_valid = false;
    guard.armed = false;
    return VRT_OK;
}

int vrt_get_scene_info(const vrt_ctx *c, vrt_scene_info *info) {
    if (!c || !info) return VRT_E_INVALID;
    *info = c->info;
    return VRT_OK;
}

int vrt_set_camera(vrt_ctx *c, const float inv_projection[16], const float inv_view[16], const float camera_pos[4]) {
    if (!c) return VRT_E_INVALID;
    if (!inv_projection || !inv_view || !camera_pos) return fail(c, VRT_E_INVALID, "vrt_set_camera: null pointer");
    std::memcpy(c->inv_proj, inv_projection, sizeof c->inv_proj);
    std::memcpy(c->inv_view, inv_view, sizeof c->inv_view);
    std::memcpy(c->cam_pos, camera_pos, sizeof c->cam_pos);
    c->have_camera = true;
    return VRT_OK;
}

int vrt_set_tile_scheduling(vrt_ctx *c, int period) {
    if (!c) return VRT_E_INVALID;
    if (period < 0) return fail(c, VRT_E_INVALID, "vrt_set_tile_scheduling: period must be >= 0");
    c->sched_period = period;
    return VRT_OK;
}

int vrt_variant_available(int variant) {
    return variant >= 0 && variant < kNumVariants && (VRT_AB || kVariantShipped[variant]) ? 1 : 0;
}

int vrt_set_variant(vrt_ctx *c, int variant) {
    if (!c) return VRT_E_INVALID;
    if (variant < 0 || variant >= kNumVariants) return fail(c, VRT_E_INVALID, "vrt_set_variant: unknown variant");
    if (!VRT_AB && !kVariantShipped[variant])
        return fail(c, VRT_E_INVALID, "vrt_set_variant: an A/B variant; this library was built without them (make AB=1)");
    c->variant = variant;
    return VRT_OK;
}


namespace {
// device images behind the host-buffer entry points: rgba8, (id, dist) and the displayed rgba8
int ensure_scratch(vrt_ctx *c, size_t px) {
    if (px <= c->scratch_pixels) return VRT_OK;
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    if (c->d_rgba) VRT_HIP(c, hipFree(c->d_rgba));
    if (c->d_id) VRT_HIP(c, hipFree(c->d_id));
    if (c->d_shown) VRT_HIP(c, hipFree(c->d_shown));
    c->d_rgba = c->d_id = c->d_shown = nullptr;
    c->scratch_pixels = 0;
    VRT_HIP(c, hipMalloc(&c->d_rgba, px * 4));
    VRT_HIP(c, hipMalloc(&c->d_id, px * 8));
    VRT_HIP(c, hipMalloc(&c->d_shown, px * 4));
    c->scratch_pixels = px;
    return VRT_OK;
}
}  // namespace

int vrt_dispatch_rows(vrt_ctx *c, int width, int height, int row_begin, int row_end, int mode, void *d_rgba8,
                      void *d_id_dist, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (row_begin < 0 || row_end > height || row_begin > row_end) return fail(c, VRT_E_INVALID, "vrt_dispatch_rows: bad row range");
    VRT_HIP(c, hipSetDevice(c->device));
    const int n = row_end - row_begin;
    return enqueue(c, width, height, row_begin, n, n > 0 ? n : 1, 0, 0, mode, d_rgba8, d_id_dist,
                   stream ? (hipStream_t)stream : c->stream);
}

int vrt_shard_rows(int height, int tile_rows, int shard, int n_shards) {
    if (height < 1 || tile_rows < 1 || n_shards < 1 || shard < 0 || shard >= n_shards) return VRT_E_INVALID;
    const int tiles = (height + tile_rows - 1) / tile_rows;
    int rows = 0;
    for (int t = shard; t < tiles; t += n_shards) {
        const int r0 = t * tile_rows;
        rows += (r0 + tile_rows <= height) ? tile_rows : height - r0;
    }
    return rows;
}

int vrt_dispatch_shard(vrt_ctx *c, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       void *d_rgba8, void *d_id_dist, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    const int rows = vrt_shard_rows(height, tile_rows, shard, n_shards);
    if (rows < 0) return fail(c, VRT_E_INVALID, "vrt_dispatch_shard: bad tile_rows/shard/n_shards");
    VRT_HIP(c, hipSetDevice(c->device));
    return enqueue(c, width, height, shard * tile_rows, rows, tile_rows, tile_rows * n_shards, 1, mode, d_rgba8,
                   d_id_dist, stream ? (hipStream_t)stream : c->stream);
}

int vrt_dispatch_tiles(vrt_ctx *c, int width, int height, int tile_rows, int shard, int n_shards, int mode, void *d_frame_rgba8,
                       void *d_frame_id_dist, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    const int rows = vrt_shard_rows(height, tile_rows, shard, n_shards);
    if (rows < 0) return fail(c, VRT_E_INVALID, "vrt_dispatch_tiles: bad tile_rows/shard/n_shards");
    VRT_HIP(c, hipSetDevice(c->device));
    // the shard's tiles at their frame rows (compact = 0): the frame may be local, a peer's, or an IPC mapping
    return enqueue(c, width, height, shard * tile_rows, rows, tile_rows, tile_rows * n_shards, 0, mode, d_frame_rgba8, d_frame_id_dist,
                   stream ? (hipStream_t)stream : c->stream);
}

int vrt_dispatch_views(vrt_ctx *c, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       const vrt_view *views, int n_views, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!views) return fail(c, VRT_E_INVALID, "vrt_dispatch_views: null views");
    const int rows = vrt_shard_rows(height, tile_rows, shard, n_shards);
    if (rows < 0) return fail(c, VRT_E_INVALID, "vrt_dispatch_views: bad tile_rows/shard/n_shards");
    VRT_HIP(c, hipSetDevice(c->device));
    return enqueue(c, width, height, shard * tile_rows, rows, tile_rows, tile_rows * n_shards, 1, mode, nullptr, nullptr,
                   stream ? (hipStream_t)stream : c->stream, views, n_views);
}

int vrt_dispatch(vrt_ctx *c, int width, int height, int mode, uint8_t *out_rgba8, int32_t *out_id_dist) {
    int r = check_frame(c, width, height);
    if (r) return r;
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * (size_t)height;
    r = ensure_scratch(c, px);
    if (r) return r;
    r = enqueue(c, width, height, 0, height, height, 0, 0, mode, out_rgba8 ? c->d_rgba : nullptr,
                out_id_dist ? c->d_id : nullptr, c->stream);
    if (r) return r;
    if (out_rgba8) VRT_HIP(c, hipMemcpyAsync(out_rgba8, c->d_rgba, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_id_dist) VRT_HIP(c, hipMemcpyAsync(out_id_dist, c->d_id, px * 8, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_dispatch_wait(vrt_ctx *c, int ticket) {
    if (!c || ticket < 0 || ticket > 1) return c ? fail(c, VRT_E_INVALID, "vrt_dispatch_wait: ticket") : VRT_E_INVALID;
    vrt_ctx::AsyncLane &ln = c->lane[ticket];
    if (!ln.busy) return VRT_OK;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipEventSynchronize(ln.done));
    ln.busy = false;
    return VRT_OK;
}

int vrt_dispatch_async(vrt_ctx *c, int width, int height, int mode, uint8_t *out_rgba8, int32_t *out_id_dist, int *ticket) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!ticket) return fail(c, VRT_E_INVALID, "vrt_dispatch_async: null ticket");
    VRT_HIP(c, hipSetDevice(c->device));
    const int k = c->next_lane;
    vrt_ctx::AsyncLane &ln = c->lane[k];
    r = vrt_dispatch_wait(c, k);   // at most two frames in flight
    if (r) return r;
    const size_t px = (size_t)width * (size_t)height;
    if (!ln.stream) {
        VRT_HIP(c, hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
        VRT_HIP(c, hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    }
    if (px > ln.pixels) {
        void *a = nullptr, *b = nullptr;
        VRT_HIP(c, hipMalloc(&a, px * 4));
        if (hipMalloc(&b, px * 8) != hipSuccess) { (void)hipFree(a); return fail(c, VRT_E_HIP, "vrt_dispatch_async: hipMalloc"); }
        (void)hipFree(ln.d_rgba);
        (void)hipFree(ln.d_id);
        ln.d_rgba = a; ln.d_id = b; ln.pixels = px;
    }
    r = enqueue(c, width, height, 0, height, height, 0, 0, mode, out_rgba8 ? ln.d_rgba : nullptr, out_id_dist ? ln.d_id : nullptr, ln.stream);
    if (r) return r;
    if (out_rgba8) VRT_HIP(c, hipMemcpyAsync(out_rgba8, ln.d_rgba, px * 4, hipMemcpyDeviceToHost, ln.stream));
    if (out_id_dist) VRT_HIP(c, hipMemcpyAsync(out_id_dist, ln.d_id, px * 8, hipMemcpyDeviceToHost, ln.stream));
    VRT_HIP(c, hipEventRecord(ln.done, ln.stream));
    ln.busy = true;
    *ticket = k;
    c->next_lane = k ^ 1;
    return VRT_OK;
}

int vrt_host_alloc(vrt_ctx *c, size_t bytes, void **host_ptr) {
    if (!c || !host_ptr || bytes == 0) return VRT_E_INVALID;
    *host_ptr = nullptr;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return VRT_OK;
}

int vrt_host_free(vrt_ctx *c, void *host_ptr) {
    if (!c) return VRT_E_INVALID;
    if (host_ptr) VRT_HIP(c, hipHostFree(host_ptr));
    return VRT_OK;
}

int vrt_dispatch_timed(vrt_ctx *c, int width, int height, int row_begin, int row_end, int mode, void *d_rgba8,
                       void *d_id_dist, void *stream, int iters, float *ms_out) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (iters < 1 || !ms_out) return fail(c, VRT_E_INVALID, "vrt_dispatch_timed: iters/ms_out");
    if (row_begin < 0 || row_end > height || row_begin >= row_end) return fail(c, VRT_E_INVALID, "vrt_dispatch_timed: bad row range");
    VRT_HIP(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    std::vector<hipEvent_t> ev((size_t)iters * 2, nullptr);
    int rc = VRT_OK;
    hipError_t he = hipSuccess;
    for (auto &e : ev)
        if (he == hipSuccess) he = hipEventCreate(&e);
    const int n = row_end - row_begin;
    for (int i = 0; i < iters && rc == VRT_OK && he == hipSuccess; ++i) {
        he = hipEventRecord(ev[2 * i], s);
        if (he == hipSuccess) rc = enqueue(c, width, height, row_begin, n, n, 0, 0, mode, d_rgba8, d_id_dist, s);
        if (he == hipSuccess && rc == VRT_OK) he = hipEventRecord(ev[2 * i + 1], s);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(s);
    for (int i = 0; i < iters && rc == VRT_OK && he == hipSuccess; ++i) he = hipEventElapsedTime(&ms_out[i], ev[2 * i], ev[2 * i + 1]);
    for (auto &e : ev)
        if (e) (void)hipEventDestroy(e);   // on every path
    if (rc == VRT_OK && he != hipSuccess) rc = fail(c, VRT_E_HIP, std::string("vrt_dispatch_timed: ") + hipGetErrorString(he));
    return rc;
}

int vrt_denoise(vrt_ctx *c, int width, int height, const void *d_rgba8, const void *d_id_dist, void *d_out_rgba8, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!d_rgba8 || !d_id_dist || !d_out_rgba8 || d_rgba8 == d_out_rgba8) return fail(c, VRT_E_INVALID, "vrt_denoise: null or aliased buffers");
    VRT_HIP(c, hipSetDevice(c->device));
    vrt::denoise::Args a;
    a.rgba = (const uint32_t *)d_rgba8;
    a.id = (const int2 *)d_id_dist;
    a.out = (uint32_t *)d_out_rgba8;
    a.width = width;
    a.height = height;
    const hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (c->denoise_variant == 1) {
        const dim3 grid((unsigned)((width + vrt::denoise::kTile - 1) / vrt::denoise::kTile),
                        (unsigned)((height + vrt::denoise::kTile - 1) / vrt::denoise::kTile));
        hipLaunchKernelGGL(vrt::denoise::denoise_kernel, grid, dim3(vrt::denoise::kTile, vrt::denoise::kTile), 0, s, a);
    } else {
        using namespace vrt::denoise;
        const dim3 grid((unsigned)((width + kTW - 1) / kTW), (unsigned)((height + 15) / 16));
        // the trace kernel's feedback scheduling, keyed as mode kSchedDenoise: one tile = one workgroup here
        a.tiles_x = (int)grid.x;
        a.n_tiles = (int)(grid.x * grid.y);
        a.group_order = nullptr;
        a.tile_cost = nullptr;
        const long groups = ((long)a.n_tiles + vrt::kGroupTiles - 1) / vrt::kGroupTiles;
        SchedState *st = nullptr;
        if (c->sched_period > 0 && groups >= kSchedMinDenoiseGroups && groups <= kSchedMaxGroups)
            st = sched_state(c, s, width, height, 0, 0, 0, kSchedDenoise, (uint32_t)a.n_tiles, (uint32_t)groups);
        if (!st) {
            hipLaunchKernelGGL((denoise_px_kernel<2, 16>), grid, dim3(kTW / 2, 16), 0, s, a);
        } else {
            const bool measure = measuring_launch(st->launches, c->sched_period);
            a.group_order = st->valid ? st->d_order : nullptr;
            if (measure) {
                a.tile_cost = st->d_cost;
                VRT_HIP(c, hipMemsetAsync(st->d_cost, 0, (size_t)groups * vrt::kGroupTiles * sizeof(uint32_t), s));
            }
            hipLaunchKernelGGL((denoise_px_kernel<2, 16, true>), dim3((unsigned)(groups * vrt::kGroupTiles)), dim3(kTW / 2, 16), 0, s, a);
            VRT_HIP(c, hipGetLastError());
            ++st->launches;
            if (measure) {
                const int rr = launch_order_kernel(c, st, s);
                if (rr) return rr;
            }
        }
    }
    VRT_HIP(c, hipGetLastError());
    return VRT_OK;
}

int vrt_denoise_host(vrt_ctx *c, int width, int height, const uint8_t *rgba8, const int32_t *id_dist, uint8_t *out_rgba8) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!rgba8 || !id_dist || !out_rgba8) return fail(c, VRT_E_INVALID, "vrt_denoise_host: null buffer");
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * (size_t)height;
    r = ensure_scratch(c, px);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(c->d_rgba, rgba8, px * 4, hipMemcpyHostToDevice, c->stream));
    VRT_HIP(c, hipMemcpyAsync(c->d_id, id_dist, px * 8, hipMemcpyHostToDevice, c->stream));
    r = vrt_denoise(c, width, height, c->d_rgba, c->d_id, c->d_shown, nullptr);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(out_rgba8, c->d_shown, px * 4, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_dispatch_frame(vrt_ctx *c, int width, int height, int mode, uint8_t *out_shown_rgba8, uint8_t *out_rgba8,
                       int32_t *out_id_dist) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!out_shown_rgba8) return fail(c, VRT_E_INVALID, "vrt_dispatch_frame: null output");
    VRT_HIP(c, hipSetDevice(c->device));
    const size_t px = (size_t)width * (size_t)height;
    r = ensure_scratch(c, px);
    if (r) return r;
    r = enqueue(c, width, height, 0, height, height, 0, 0, mode, c->d_rgba, c->d_id, c->stream);
    if (r) return r;
    r = vrt_denoise(c, width, height, c->d_rgba, c->d_id, c->d_shown, nullptr);
    if (r) return r;
    VRT_HIP(c, hipMemcpyAsync(out_shown_rgba8, c->d_shown, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_rgba8) VRT_HIP(c, hipMemcpyAsync(out_rgba8, c->d_rgba, px * 4, hipMemcpyDeviceToHost, c->stream));
    if (out_id_dist) VRT_HIP(c, hipMemcpyAsync(out_id_dist, c->d_id, px * 8, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_synchronize(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

int vrt_set_profiling(vrt_ctx *c, int max_launches) {
    if (!c) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(c->device));
    c->prof_count = 0;
    c->prof_seen = 0;
    c->profiling = max_launches > 0;
    while (c->profiling && c->prof_events.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        VRT_HIP(c, hipEventCreate(&e));
        c->prof_events.push_back(e);
    }
    return VRT_OK;
}

int vrt_set_profiling_stride(vrt_ctx *c, int every) {
    if (!c || every < 1) return VRT_E_INVALID;
    c->prof_stride = (size_t)every;
    return VRT_OK;
}

int vrt_profile_read(vrt_ctx *c, float *ms_out, int cap) {
    if (!c || !ms_out || cap < 0) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(c->device));
    int n = 0;
    for (size_t i = 0; i < c->prof_count && n < cap; ++i, ++n) {
        VRT_HIP(c, hipEventSynchronize(c->prof_events[2 * i + 1]));
        VRT_HIP(c, hipEventElapsedTime(&ms_out[n], c->prof_events[2 * i], c->prof_events[2 * i + 1]));
    }
    c->prof_count = 0;
    return n;
}

void *vrt_stream(vrt_ctx *c) { return c ? (void *)c->stream : nullptr; }
int vrt_device(const vrt_ctx *c) { return c ? c->device : VRT_E_INVALID; }

// Arithmetic-contract probe (see math_probe_kernel): host arrays in/out, synchronous.
// A/B switch (tests, tools): 0 runs the full path tracer as the one kernel of round 1, 1 (default) as two (vrt_bounce.hip.h)
int vrt_debug_set_full_split(vrt_ctx *c, int on) {
    if (!c) return VRT_E_INVALID;
    if (on && !VRT_AB) return fail(c, VRT_E_INVALID, "vrt_debug_set_full_split: the two-kernel form exists in A/B builds only (make AB=1)");
    c->full_split = on != 0;
    return VRT_OK;
}

// Host-only (tests): the ray-generation table the dispatcher would build for this inverse projection and frame shape.
// Returns 1 and fills out_x[width], out_y[height], out_z when the projection has a table, 0 when it has none.
int vrt_debug_ray_table(const float inv_projection[16], int width, int height, float *out_x, float *out_y, float *out_z) {
    if (!inv_projection || width < 1 || height < 1 || !out_x || !out_y || !out_z) return VRT_E_INVALID;
    std::vector<float> tab;
    float z = 0.0f;
    if (!build_ray_table(inv_projection, width, height, tab, z)) return 0;
    std::memcpy(out_x, tab.data(), (size_t)width * sizeof(float));
    std::memcpy(out_y, tab.data() + width, (size_t)height * sizeof(float));
    *out_z = z;
    return 1;
}
// ... and whether an inverse view matrix keeps the second normalisation in range (view_matrix_in_range())
int vrt_debug_view_in_range(const float inv_view[16]) { return inv_view ? (view_matrix_in_range(inv_view) ? 1 : 0) : VRT_E_INVALID; }

// A/B switch (tests, tools): 0 makes every launch run the shader's own ray-generation prologue, 1 (default) lets views
// whose projection allows it read the per-column / per-row tables (ray_table())
// Host-only (tests): what the dispatcher would tell the kernels about wide root 0 for this tree, these world bounds and
// this eye: out[0] = root0_only, out[1] = log2 of the chosen root's side, out[2..4] = its minimum corner, out[5] = log2 of
// the side of build_wide()'s root. Returns 0, VRT_E_MALFORMED, or VRT_E_STATE when the scene has no wide form.
int vrt_debug_root0(const uint8_t *texels, size_t used_bytes, const int32_t wmin[3], const int32_t wmax[3], const int32_t eye[3],
                    int32_t out[6]) {
    if (!wmin || !wmax || !eye || !out) return VRT_E_INVALID;
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    vrt::WideTree wt;
    if (vrt::has_unit_internal_node(lay.records, wmin, wmax) || !vrt::build_wide(lay.records, wmin, wmax, wt, err) || wt.roots.empty())
        return VRT_E_STATE;
    uint32_t node = wt.roots[0].node;
    int shift = wt.roots[0].shift, mn[3] = {wt.roots[0].origin[0], wt.roots[0].origin[1], wt.roots[0].origin[2]};
    out[5] = shift;
    out[0] = vrt::content_only_in_root0(lay.records, wt) ? 1 : 0;
    const int eyes[1][3] = {{eye[0], eye[1], eye[2]}};
    if (out[0]) vrt::tighten_root0(wt, eyes, 1, vrt::v3::kAnchorShift, node, shift, mn);
    out[1] = shift; out[2] = mn[0]; out[3] = mn[1]; out[4] = mn[2];
    return VRT_OK;
}

// A/B switch (tests, tools): 0 keeps KArgs::root0_only off (rays that leave wide root 0 walk the empty octants' records)
int vrt_debug_set_root0_only(vrt_ctx *c, int on) {
    if (!c) return VRT_E_INVALID;
    c->root0_only_on = on != 0;
    c->tight_root_on = on == 1;   // 2: the shortcut with wide root 0 as build_wide() found it
    return VRT_OK;
}

int vrt_debug_set_ray_tables(vrt_ctx *c, int on) {
    if (!c) return VRT_E_INVALID;
    c->ray_tables_on = on != 0;
    return VRT_OK;
}

int vrt_debug_set_bounce(vrt_ctx *c, int refill_below, int waves_per_simd) {
    if (!c || refill_below < 1 || refill_below > 65 || waves_per_simd < 1 || waves_per_simd > 8) return VRT_E_INVALID;
    c->bounce_refill_below = refill_below;
    c->bounce_waves_per_simd = waves_per_simd;
    return VRT_OK;
}

int vrt_debug_set_denoise_variant(vrt_ctx *c, int v) {
    if (!c || v < 0 || v > 1) return VRT_E_INVALID;
    c->denoise_variant = v;
    return VRT_OK;
}

// Experiment hook (tools/tile_order_ab.py): caller-owned device buffers -- a workgroup permutation and/or a per-tile
// tick buffer -- for the default kernel instead of the scheduler's own (KArgs::group_order / tile_cost). enable = 0
// hands the launches back to the scheduler.
int vrt_debug_set_tile_order(vrt_ctx *c, int enable, const void *d_group_order, void *d_tile_cost) {
    if (!c) return VRT_E_INVALID;
    c->dbg_sched = enable != 0;
    c->dbg_group_order = enable ? (const uint32_t *)d_group_order : nullptr;
    c->dbg_tile_cost = enable ? (uint32_t *)d_tile_cost : nullptr;
    return VRT_OK;
}

// Reads back the scheduler's current workgroup order for the shape last launched on `stream` (tests): returns the
// number of workgroups (0 when no order has been derived yet), at most `cap` entries copied.
long vrt_debug_sched_order(vrt_ctx *c, void *stream, uint32_t *out, size_t cap) {
    if (!c) return VRT_E_INVALID;
    const SchedState *best = nullptr;
    for (const SchedState &st : c->sched)
        if (st.stream == (stream ? (hipStream_t)stream : c->stream) && (!best || st.last_use > best->last_use)) best = &st;
    if (!best || !best->valid) return 0;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipStreamSynchronize(best->stream));
    const size_t n = best->n_groups < cap ? best->n_groups : cap;
    if (out && n) VRT_HIP(c, hipMemcpy(out, best->d_order, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (long)best->n_groups;
}

int vrt_debug_math(vrt_ctx *c, int op, const float *x, const float *y, float *out, int n) {
    if (!c || !x || !y || !out || n < 1) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(c->device));
    float *dx = nullptr, *dy = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * sizeof(float);
    struct Free { float *&a, *&b, *&o; ~Free() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(o); } } free_on_exit{dx, dy, dout};
    VRT_HIP(c, hipMalloc((void **)&dx, bytes));
    VRT_HIP(c, hipMalloc((void **)&dy, bytes));
    VRT_HIP(c, hipMalloc((void **)&dout, bytes));
    VRT_HIP(c, hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, c->stream));
    VRT_HIP(c, hipMemcpyAsync(dy, y, bytes, hipMemcpyHostToDevice, c->stream));
    if (op >= 10 && op < 30)
        hipLaunchKernelGGL(vrt::full::math_probe_full_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, dx, dy, dout, n);
    else
        hipLaunchKernelGGL(vrt::math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, op, dx, dy, dout, n);
    VRT_HIP(c, hipGetLastError());
    VRT_HIP(c, hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

// Host-only check of the wide layout (tests without a GPU): builds it for the texel stream and world
// bounds and answers n point queries through it. out: n * 8 words = w0, w1, mn[3], mx[3].
// stats (optional): wide nodes, roots. Returns 0, VRT_E_MALFORMED, or VRT_E_STATE when the scene has no wide form.
int vrt_debug_wide_find(const uint8_t *texels, size_t used_bytes, const int32_t wmin[3], const int32_t wmax[3],
                        const int32_t *points, size_t n, uint32_t *out, uint32_t *stats) {
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    vrt::WideTree wt;
    if (!vrt::build_wide(lay.records, wmin, wmax, wt, err)) return VRT_E_STATE;
    if (stats) { stats[0] = wt.n_nodes; stats[1] = (uint32_t)wt.roots.size(); }
    for (size_t i = 0; i < n; ++i) {
        uint32_t w0, w1;
        int mn[3], mx[3];
        (void)vrt::wide_find_host(lay.records, wt, wmin, wmax, points + 3 * i, w0, w1, mn, mx);
        uint32_t *o = out + 8 * i;
        o[0] = w0; o[1] = w1;
        for (int k = 0; k < 3; ++k) { o[2 + k] = (uint32_t)mn[k]; o[5 + k] = (uint32_t)mx[k]; }
    }
    return VRT_OK;
}

// Host-only check of the edit patch (no device): lays out the tree before and after an edit of voxel (x, y, z) from
// their texel streams, patches the "before" structures with the sub-tree taken from the "after" ones, and counts the
// query points whose lookup (leaf words + node box), through the wide layout and through the records alone, differs
// between the patched and the freshly built structures. info: [0] depth of the node replaced (0: no patchable
// ancestor, nothing compared), [1] records appended, [2] wide cells appended, [3] 1 when the stream's texel count
// tracked by the patch equals the "after" stream's. sparse: the sub-tree carries only the nodes that contain the
// voxel, everything else as kKeep records. Returns the number of differing points or a negative code.
long vrt_debug_patch_check(const uint8_t *before, size_t before_bytes, const uint8_t *after, size_t after_bytes,
                           const int32_t wmin[3], const int32_t wmax[3], int x, int y, int z, const int32_t *points, size_t n,
                           uint32_t *info, int sparse) {
    vrt::Layout lb, la;
    std::string err;
    if (!vrt::build_layout(before, before_bytes, lb, err) || !vrt::build_layout(after, after_bytes, la, err)) return VRT_E_MALFORMED;
    vrt::WideTree wb, wa;
    const bool wide_b = !vrt::has_unit_internal_node(lb.records, wmin, wmax) && vrt::build_wide(lb.records, wmin, wmax, wb, err);
    const bool wide_a = !vrt::has_unit_internal_node(la.records, wmin, wmax) && vrt::build_wide(la.records, wmin, wmax, wa, err);
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    const int p[3] = {x, y, z};
    vrt::PatchSite site;
    std::vector<vrt::Record> sub;
    int max_depth = 15;
    bool have = false;
    while (max_depth >= 1 && vrt::plan_patch(lb.records, wb, wide_b, wmin, wmax, p, max_depth, site)) {
        if (vrt::extract_subtree(la.records, site.path, site.depth, sub, sparse ? p : nullptr, wmin, wmax)) { have = true; break; }
        max_depth = site.depth - 1;
    }
    if (!have) return 0;
    const size_t n_rec = lb.records.size(), n_cells = wb.cells.size();
    vrt::PatchRanges rg;
    if (!vrt::apply_patch(lb.records, wb, wide_b, site, sub.data(), sub.size(), rg, err)) return VRT_E_MALFORMED;
    bool wide_p = wide_b;
    if (wide_b && rg.wide_invalid) wide_p = !vrt::has_unit_internal_node(lb.records, wmin, wmax) && vrt::build_wide(lb.records, wmin, wmax, wb, err);
    if (info) {
        info[0] = (uint32_t)site.depth;
        info[1] = (uint32_t)(lb.records.size() - n_rec);
        info[2] = (uint32_t)(wb.cells.size() > n_cells ? wb.cells.size() - n_cells : 0);
        info[3] = ((long)(before_bytes / 4) + rg.texel_delta == (long)(after_bytes / 4)) ? 1u : 0u;
    }
    if (wide_p != wide_a) return VRT_E_STATE;
    const vrt::WideTree none;
    long bad = 0;
    for (size_t i = 0; i < n; ++i) {
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 0 && !wide_p) continue;
            uint32_t a0, a1, b0, b1;
            int amn[3], amx[3], bmn[3], bmx[3];
            const int ka = vrt::wide_find_host(lb.records, pass == 0 ? wb : none, wmin, wmax, points + 3 * i, a0, a1, amn, amx);
            const int kb = vrt::wide_find_host(la.records, pass == 0 ? wa : none, wmin, wmax, points + 3 * i, b0, b1, bmn, bmx);
            bool same = ka == kb && a0 == b0 && a1 == b1;
            for (int k = 0; k < 3; ++k) same = same && amn[k] == bmn[k] && amx[k] == bmx[k];
            if (!same) { ++bad; break; }
        }
    }
    return bad;
}

// Host-only view of the device layout for tests that run without a GPU:
// writes up to cap records (8 bytes each) and returns the record count, or <0.
long vrt_debug_build_layout(const uint8_t *texels, size_t used_bytes, uint32_t *records_out, size_t cap_records,
                            vrt_scene_info *info) {
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return VRT_E_MALFORMED;
    if (records_out)
        for (size_t i = 0; i < lay.records.size() && i < cap_records; ++i) {
            records_out[2 * i] = lay.records[i].w0;
            records_out[2 * i + 1] = lay.records[i].w1;
        }
    if (info) {
        std::memset(info, 0, sizeof *info);
        info->n_texels = (uint32_t)(used_bytes / 4);
        info->n_records = (uint32_t)lay.records.size();
        info->n_internal = lay.n_internal;
        info->n_leaves = lay.n_leaves;
        info->max_depth = lay.max_depth;
    }
    return (long)lay.records.size();
}

}  // extern "C"
