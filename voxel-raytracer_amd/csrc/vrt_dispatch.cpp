// vrt_dispatch.cpp -- the dispatch boundary (src/main.cpp:941-946): builds the kernel arguments of one launch, chooses the kernel
// variant the scene and the view allow, runs the feedback tile scheduler and the per-projection ray tables, and carries the
// vrt_dispatch* entry points of include/vrt.h. Host code; the kernels are behind vrt_launch.h.
#include "vrt_internal.h"

#include <cmath>
#include <cstdio>
#include <new>

using namespace vrt_internal;
#include "vrt_launch.h"

namespace vrt_internal {

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
        hipMalloc((void **)&slot->d_order, ((size_t)n_groups + 1) * sizeof(uint32_t)) != hipSuccess ||   // + KArgs::split_count
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
bool measuring_launch(uint64_t launches, int period) {
    return period == 1 || launches % (uint64_t)period == 1;
}

// After a measuring launch, on the same stream: reads that launch's ticks, rewrites the order the next launches read.
int launch_order_kernel(vrt_ctx *c, SchedState *st, hipStream_t s) {
    bool &raised = c->order_lds_raised;   // per context, i.e. per device: the attribute does not carry over to another one
    const bool raise = (size_t)st->n_groups * sizeof(uint32_t) > 48 * 1024 && !raised;
    // the general full path tracer is built for five waves per SIMD; the other states never read the split count
    const uint32_t wave_slots = st->mode == VRT_MODE_FULL ? (uint32_t)c->n_cus * 4u * 5u : 0u;
    VRT_HIP(c, vrt::launch::tile_order(st->d_cost, st->n_groups, st->d_order, wave_slots, raise, (size_t)kSchedMaxGroups * sizeof(uint32_t), s));
    if (raise) raised = true;
    st->valid = true;
    return VRT_OK;
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

// views == nullptr: one view, the context's camera (vrt_set_camera) rendering into d_rgba / d_id.
int enqueue(vrt_ctx *c, int width, int height, int row0, int n_rows, int tile_rows, int row_stride, int compact,
            int mode, void *d_rgba, void *d_id, hipStream_t s, const vrt_view *views, int n_views) {
    if (!c->have_scene) return vrt_fail(c, VRT_E_STATE, "vrt_dispatch: no octree uploaded (call vrt_upload_octree first)");
    if (c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_dispatch: a patch batch is open (call vrt_patch_end first)");
    if (!views && !c->have_camera) return vrt_fail(c, VRT_E_STATE, "vrt_dispatch: no camera set (call vrt_set_camera first)");
    if (n_views < 1 || n_views > vrt::kMaxViews) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_views: 1 to 4 views per launch");
    if (mode != VRT_MODE_PRIMARY && mode != VRT_MODE_PRIMARY_SHADOW && mode != VRT_MODE_FULL)
        return vrt_fail(c, VRT_E_INVALID, "unknown mode");
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
    a.split_count = nullptr;
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
    // the full path tracer as two tile-coherent passes, where the scene and the view allow it
    bool two_pass = false;
    if (mode == VRT_MODE_FULL && c->two_pass_on && v.trav == 4 && n_views == 1 && c->variant == 0 && !split && vs.v[0].out_rgba) {
        if (!c->scene_opaque_valid) { c->scene_opaque = vrt::tree_is_opaque(c->host_records); c->scene_opaque_valid = true; }
        const uint32_t eye_alpha = vs.v[0].eye0 >> 24, eye_b = vs.v[0].eye1 & 0xffu;
        two_pass = c->scene_opaque && eye_alpha == 0u && (eye_b == 0u || eye_b == 85u || eye_b == 255u);
    }
    // the general full path tracer starts the heaviest groups of an ordered, non-measuring launch as part-tile waves (KArgs::split_count)
    if (mode == VRT_MODE_FULL && !two_pass && !split && st && a.group_order && v.block == 64 && c->heavy_split_on) {
        a.split_count = st->d_order + st->n_groups;
        grid += (long)vrt::kSplitMaxGroups * vrt::kGroupTiles * (vrt::kSplitParts - 1);
        // a measuring launch: the part-tile waves of a tile meet in its ticks with atomicMax
        if (a.tile_cost) VRT_HIP(c, hipMemsetAsync(st->d_cost, 0, (size_t)st->n_groups * vrt::kGroupTiles * sizeof(uint32_t), s));
    }
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
    if (two_pass && c->two_pass_form >= 5) {
        e = vrt::launch::trace_full_opaque(a, vs, (int)grid, c->two_pass_form, s, ev0, ev1);
    } else if (two_pass) {
        vrt_ctx::SeedBuffer *sb = nullptr;
        for (auto &b : c->seeds)
            if (b.stream == s) sb = &b;
        if (!sb) {
            if (c->seeds.size() < 8) {
                c->seeds.emplace_back();
                sb = &c->seeds.back();
            } else {
                for (auto &b : c->seeds)
                    if (!sb || b.last_use < sb->last_use) sb = &b;
                VRT_HIP(c, hipStreamSynchronize(sb->stream));   // its launches may still be in flight there
            }
            sb->stream = s;
        }
        const size_t need = (size_t)(a.group_order ? groups * vrt::kGroupTiles : tiles);
        if (need > sb->tiles) {
            VRT_HIP(c, hipStreamSynchronize(s));
            uint32_t *fresh = nullptr;
            VRT_HIP(c, hipMalloc((void **)&fresh, need * vrt::kSeedPlanesHost * 64 * sizeof(uint32_t)));
            if (sb->d) (void)hipFree(sb->d);
            sb->d = fresh;
            sb->tiles = need;
        }
        sb->last_use = ++c->seed_tick;
        a.defer_rec = reinterpret_cast<float *>(sb->d);
        e = vrt::launch::trace_full_two_pass(a, vs, (int)grid, s, ev0, ev1);
    } else
#if VRT_AB
    if (split) e = vrt::launch::trace_split(a, vs, (int)grid, c->n_cus * 4 * c->bounce_waves_per_simd, c->bounce_refill_below, s, ev0, ev1);
    else
#endif
    e = vrt::launch::trace(mode, v, a, vs, (int)grid, lds_bytes, s, ev0, ev1);
    if (e != hipSuccess) return vrt_fail(c, VRT_E_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
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

}  // namespace vrt_internal

extern "C" {

int vrt_dispatch_rows(vrt_ctx *c, int width, int height, int row_begin, int row_end, int mode, void *d_rgba8,
                      void *d_id_dist, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (row_begin < 0 || row_end > height || row_begin > row_end) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_rows: bad row range");
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
    if (rows < 0) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_shard: bad tile_rows/shard/n_shards");
    VRT_HIP(c, hipSetDevice(c->device));
    return enqueue(c, width, height, shard * tile_rows, rows, tile_rows, tile_rows * n_shards, 1, mode, d_rgba8,
                   d_id_dist, stream ? (hipStream_t)stream : c->stream);
}

int vrt_dispatch_tiles(vrt_ctx *c, int width, int height, int tile_rows, int shard, int n_shards, int mode, void *d_frame_rgba8,
                       void *d_frame_id_dist, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    const int rows = vrt_shard_rows(height, tile_rows, shard, n_shards);
    if (rows < 0) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_tiles: bad tile_rows/shard/n_shards");
    VRT_HIP(c, hipSetDevice(c->device));
    // the shard's tiles at their frame rows (compact = 0): the frame may be local, a peer's, or an IPC mapping
    return enqueue(c, width, height, shard * tile_rows, rows, tile_rows, tile_rows * n_shards, 0, mode, d_frame_rgba8, d_frame_id_dist,
                   stream ? (hipStream_t)stream : c->stream);
}

int vrt_dispatch_views(vrt_ctx *c, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       const vrt_view *views, int n_views, void *stream) {
    int r = check_frame(c, width, height);
    if (r) return r;
    if (!views) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_views: null views");
    const int rows = vrt_shard_rows(height, tile_rows, shard, n_shards);
    if (rows < 0) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_views: bad tile_rows/shard/n_shards");
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
    if (!c || ticket < 0 || ticket > 1) return c ? vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_wait: ticket") : VRT_E_INVALID;
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
    if (!ticket) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_async: null ticket");
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
        if (hipMalloc(&b, px * 8) != hipSuccess) { (void)hipFree(a); return vrt_fail(c, VRT_E_HIP, "vrt_dispatch_async: hipMalloc"); }
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
    if (iters < 1 || !ms_out) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_timed: iters/ms_out");
    if (row_begin < 0 || row_end > height || row_begin >= row_end) return vrt_fail(c, VRT_E_INVALID, "vrt_dispatch_timed: bad row range");
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
    if (rc == VRT_OK && he != hipSuccess) rc = vrt_fail(c, VRT_E_HIP, std::string("vrt_dispatch_timed: ") + hipGetErrorString(he));
    return rc;
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

int vrt_set_tile_scheduling(vrt_ctx *c, int period) {
    if (!c) return VRT_E_INVALID;
    if (period < 0) return vrt_fail(c, VRT_E_INVALID, "vrt_set_tile_scheduling: period must be >= 0");
    c->sched_period = period;
    return VRT_OK;
}


// ---- documented switches (include/vrt.h): pixels never depend on them ----------------------------------------------------------
int vrt_set_option(vrt_ctx *c, int option, int value) {
    if (!c) return VRT_E_INVALID;
    switch (option) {
        case VRT_OPT_RAY_TABLES:
            if (value != 0 && value != 1) break;
            c->ray_tables_on = value != 0;
            return VRT_OK;
        case VRT_OPT_EMPTY_OCTANTS:
            if (value < 0 || value > 2) break;
            c->root0_only_on = value != 0;
            c->tight_root_on = value == 1;   // 2: the shortcut with wide root 0 as build_wide() found it
            return VRT_OK;
        case VRT_OPT_FULL_OPAQUE:
            if (value != 0 && value != 1 && (value < 5 || value > 7)) break;
            c->two_pass_on = value != 0;
            c->two_pass_form = value;   // 1: two kernels and a seed buffer; 5, 6, 7: both stages in one kernel at that many waves per SIMD
            return VRT_OK;
        case VRT_OPT_HEAVY_TILES:
            if (value != 0 && value != 1) break;
            c->heavy_split_on = value != 0;
            return VRT_OK;
        case VRT_OPT_DISPLAY_KERNEL:
            if (value == 0 || value == 2 || value == 3 || (value == 1 && VRT_AB)) { c->denoise_variant = value; return VRT_OK; }
            return vrt_fail(c, VRT_E_INVALID, "vrt_set_option: the one-pixel-per-lane display kernel exists in A/B builds only (make AB=1)");
        default:
            return vrt_fail(c, VRT_E_INVALID, "vrt_set_option: unknown option");
    }
    return vrt_fail(c, VRT_E_INVALID, "vrt_set_option: value out of range");
}

int vrt_set_tile_order(vrt_ctx *c, int enable, const void *d_group_order, void *d_tile_cost) {
    if (!c) return VRT_E_INVALID;
    c->dbg_sched = enable != 0;
    c->dbg_group_order = enable ? (const uint32_t *)d_group_order : nullptr;
    c->dbg_tile_cost = enable ? (uint32_t *)d_tile_cost : nullptr;
    return VRT_OK;
}

long vrt_get_tile_order(vrt_ctx *c, void *stream, uint32_t *out, size_t cap) {
    if (!c) return VRT_E_INVALID;
    const SchedState *best = nullptr;
    for (const SchedState &st : c->sched)
        if (st.stream == (stream ? (hipStream_t)stream : c->stream) && (!best || st.last_use > best->last_use)) best = &st;
    if (!best || !best->valid) return 0;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipStreamSynchronize(best->stream));
    const size_t n = (size_t)best->n_groups + 1 < cap ? (size_t)best->n_groups + 1 : cap;   // the order, then the split count (KArgs::split_count)
    if (out && n) VRT_HIP(c, hipMemcpy(out, best->d_order, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return (long)best->n_groups;
}

#if VRT_AB
// A/B builds only (tools/full_split_ab.py): the full path tracer as two kernels with cross-wave repacking (ab/vrt_bounce.hip.h)
int vrt_ab_set_full_split(vrt_ctx *c, int on) {
    if (!c) return VRT_E_INVALID;
    c->full_split = on != 0;
    return VRT_OK;
}
int vrt_ab_set_bounce(vrt_ctx *c, int refill_below, int waves_per_simd) {
    if (!c || refill_below < 1 || refill_below > 65 || waves_per_simd < 1 || waves_per_simd > 8) return VRT_E_INVALID;
    c->bounce_refill_below = refill_below;
    c->bounce_waves_per_simd = waves_per_simd;
    return VRT_OK;
}
#endif

}  // extern "C"
