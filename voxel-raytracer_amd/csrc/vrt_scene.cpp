// vrt_scene.cpp -- the context and what it holds: create / destroy (GL object setup and teardown, src/main.cpp:432-474, 973-983),
// the scalar uniforms (:689-695, 932-938), the camera block (:643-656), the octree upload (updateGPUTexture, :264-311) and the
// layouts derived from it on the device.
#include "vrt_internal.h"

#include <cmath>
#include <cstdio>
#include <new>

using namespace vrt_internal;
#include "vrt_launch.h"

namespace {
thread_local std::string g_create_error;
}

namespace vrt_internal {

int check_frame(vrt_ctx *c, int width, int height) {
    if (!c) return VRT_E_INVALID;
    if (width < 1 || height < 1 || (long)width * (long)height > (1L << 30))
        return vrt_fail(c, VRT_E_INVALID, "width/height out of range");
    return VRT_OK;
}

// src/main.cpp:266-268: tex_dim = (size_t)ceil(cbrt((double)total_texels)), at least 1
uint32_t dim_of_texels(size_t texels) {
    const size_t d = (size_t)ceil(cbrt((double)texels));
    return (uint32_t)(d == 0 ? 1 : d);
}

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
    if (from + n > c->wide.cells.size() || from + n > c->cells_capacity) return vrt_fail(c, VRT_E_STATE, "upload_cells: range outside the wide layout");
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

}  // namespace vrt_internal

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
            if ((e = vrt::launch::kernarg_probe(a, vs, d_bad, c->stream)) == hipSuccess && (e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, c->stream)) == hipSuccess)
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
    if (!c->seeds.empty()) (void)hipDeviceSynchronize();   // their launches may be on the caller's streams
    for (auto &b : c->seeds) (void)hipFree(b.d);
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
    if (!c || !p) return c ? vrt_fail(c, VRT_E_INVALID, "vrt_set_params: null params") : VRT_E_INVALID;
    for (int i = 0; i < 3; ++i)
        if (p->world_max[i] < p->world_min[i]) return vrt_fail(c, VRT_E_INVALID, "vrt_set_params: world_max < world_min");
    bool bounds_change = false;
    for (int i = 0; i < 3; ++i)
        bounds_change = bounds_change || p->world_min[i] != c->params.world_min[i] || p->world_max[i] != c->params.world_max[i];
    // an open batch holds indices into the layouts the world bounds shaped (records_before, cells_before, repointed cells)
    if (bounds_change && c->batch.open)
        return vrt_fail(c, VRT_E_STATE, "vrt_set_params: the world bounds cannot change while a patch batch is open (call vrt_patch_end first)");
    c->params = *p;
    if (bounds_change) c->analysis_valid = false;
    return VRT_OK;
}

int vrt_upload_octree(vrt_ctx *c, const uint8_t *texels, size_t used_bytes, uint32_t tex_dim) {
    if (!c) return VRT_E_INVALID;
    if (c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_upload_octree: a patch batch is open (call vrt_patch_end first)");
    if (used_bytes % 4 != 0) return vrt_fail(c, VRT_E_INVALID, "vrt_upload_octree: used_bytes must be a multiple of 4");
    if (used_bytes / 4 > (1u << 23)) return vrt_fail(c, VRT_E_MALFORMED, "vrt_upload_octree: more than 2^23 texels cannot be addressed by 23-bit node pointers");
    if (tex_dim == 0) tex_dim = 1;
    vrt::Layout lay;
    std::string err;
    if (!vrt::build_layout(texels, used_bytes, lay, err)) return vrt_fail(c, VRT_E_MALFORMED, "vrt_upload_octree: " + err);
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
    c->scene_opaque_valid = false;
    c->have_scene = true;
    return VRT_OK;
}

// Extension beyond the reference boundary: take the device record array (vrt_layout.h) directly, as
// libvrt_host.so emits it from the pointer octree (vrth_world_records). Skips the texel stream, and with
// it the stream's 23-bit pointer limit and the flatten + re-parse on every edit.
int vrt_upload_records(vrt_ctx *c, const uint32_t *records, size_t n_records, uint32_t tex_dim) {
    if (!c) return VRT_E_INVALID;
    if (c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_upload_records: a patch batch is open (call vrt_patch_end first)");
    if (!records || n_records == 0 || n_records > (1ull << 31)) return vrt_fail(c, VRT_E_INVALID, "vrt_upload_records: bad record array");
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
        if (base <= i || (size_t)base + n_child > n_records) return vrt_fail(c, VRT_E_MALFORMED, "vrt_upload_records: child index out of order or range");
        uint32_t rank = 0;
        for (uint32_t ci = 0; ci < 8; ++ci) {
            if (!((mask >> ci) & 1u)) continue;
            const size_t idx = (size_t)base + rank++;
            if (kind[idx] != 0) return vrt_fail(c, VRT_E_MALFORMED, "vrt_upload_records: a record has two parents");
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
    c->scene_opaque_valid = false;
    c->have_scene = true;
    return VRT_OK;
}

int vrt_get_scene_info(const vrt_ctx *c, vrt_scene_info *info) {
    if (!c || !info) return VRT_E_INVALID;
    *info = c->info;
    return VRT_OK;
}

int vrt_set_camera(vrt_ctx *c, const float inv_projection[16], const float inv_view[16], const float camera_pos[4]) {
    if (!c) return VRT_E_INVALID;
    if (!inv_projection || !inv_view || !camera_pos) return vrt_fail(c, VRT_E_INVALID, "vrt_set_camera: null pointer");
    std::memcpy(c->inv_proj, inv_projection, sizeof c->inv_proj);
    std::memcpy(c->inv_view, inv_view, sizeof c->inv_view);
    std::memcpy(c->cam_pos, camera_pos, sizeof c->cam_pos);
    c->have_camera = true;
    return VRT_OK;
}

int vrt_variant_available(int variant) {
    return variant >= 0 && variant < kNumVariants && (VRT_AB || kVariantShipped[variant]) ? 1 : 0;
}

int vrt_set_variant(vrt_ctx *c, int variant) {
    if (!c) return VRT_E_INVALID;
    if (variant < 0 || variant >= kNumVariants) return vrt_fail(c, VRT_E_INVALID, "vrt_set_variant: unknown variant");
    if (!VRT_AB && !kVariantShipped[variant])
        return vrt_fail(c, VRT_E_INVALID, "vrt_set_variant: an A/B variant; this library was built without them (make AB=1)");
    c->variant = variant;
    return VRT_OK;
}

int vrt_synchronize(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipStreamSynchronize(c->stream));
    return VRT_OK;
}

void *vrt_stream(vrt_ctx *c) { return c ? (void *)c->stream : nullptr; }
int vrt_device(const vrt_ctx *c) { return c ? c->device : VRT_E_INVALID; }

}  // extern "C"
