// vrt_patch.cpp -- EXTENSION: edits without re-flattening and re-uploading the tree (the reference does both on every build /
// destroy click, src/main.cpp:843-914): patch plan / apply, batches, compaction. See include/vrt.h.
#include "vrt_internal.h"

#include <cmath>
#include <cstdio>
#include <new>

using namespace vrt_internal;

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
        if (ci > 7) return vrt_fail(c, VRT_E_INVALID, "vrt_patch_apply: bad path");
        for (int k = 0; k < 3; ++k) {
            const int mid = lo[k] + ((hi[k] - lo[k]) >> 1);
            if ((ci >> (2 - k)) & 1u) lo[k] = mid; else hi[k] = mid;
        }
    }
    // a patch that voided the wide layout earlier in the batch leaves the rest to the record array alone
    const bool wide_now = c->wide_ok && !bt.wide_invalid;
    if (!vrt::plan_patch(c->host_records, c->wide, wide_now, c->params.world_min, c->params.world_max, lo, patch.depth, site) ||
        site.depth != patch.depth || std::memcmp(site.path, patch.path, (size_t)patch.depth) != 0)
        return vrt_fail(c, VRT_E_STATE, "vrt_patch_apply: the path does not name a patchable node of the uploaded tree");
    vrt::PatchRanges rg;
    std::string why;
    if (!vrt::apply_patch(c->host_records, c->wide, wide_now, site, reinterpret_cast<const vrt::Record *>(subtree_records), n_records, rg, why))
        return vrt_fail(c, VRT_E_MALFORMED, "vrt_patch_apply: " + why);   // apply_patch modifies nothing when it refuses
    bt.dirty = true;
    c->scene_opaque_valid = false;   // the tree changed: what the two-pass path tracer may assume about it is re-derived
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

extern "C" {

int vrt_patch_plan_box(vrt_ctx *c, const int32_t lo[3], const int32_t hi[3], int max_depth, vrt_patch *out) {
    if (!c || !out || !lo || !hi) return c ? vrt_fail(c, VRT_E_INVALID, "vrt_patch_plan: null argument") : VRT_E_INVALID;
    if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return vrt_fail(c, VRT_E_INVALID, "vrt_patch_plan_box: lo > hi");
    if (!c->have_scene) return vrt_fail(c, VRT_E_STATE, "vrt_patch_plan: no octree uploaded");
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
    // the deepest patchable ancestor of the box's first voxel whose cube also holds its last one
    const int p[3] = {lo[0], lo[1], lo[2]};
    int md = max_depth > 15 ? 15 : max_depth;
    while (md >= 1) {
        vrt::PatchSite site;
        if (!vrt::plan_patch(c->host_records, c->wide, c->wide_ok && !c->batch.wide_invalid, c->params.world_min, c->params.world_max, p, md, site))
            break;
        int mn[3], mx[3];
        for (int k = 0; k < 3; ++k) { mn[k] = c->params.world_min[k]; mx[k] = c->params.world_max[k]; }
        for (int d = 0; d < site.depth; ++d)
            for (int k = 0; k < 3; ++k) {
                const int mid = mn[k] + ((mx[k] - mn[k]) >> 1);
                if ((site.path[d] >> (2 - k)) & 1u) mn[k] = mid; else mx[k] = mid;
            }
        if (hi[0] < mx[0] && hi[1] < mx[1] && hi[2] < mx[2] && hi[0] >= mn[0] && hi[1] >= mn[1] && hi[2] >= mn[2]) {
            out->depth = site.depth;
            std::memcpy(out->path, site.path, sizeof out->path);
            return VRT_OK;
        }
        md = site.depth - 1;
    }
    return vrt_fail(c, VRT_E_STATE, "vrt_patch_plan: no patchable ancestor (full upload needed)");
}

int vrt_patch_plan(vrt_ctx *c, int x, int y, int z, int max_depth, vrt_patch *out) {
    const int32_t p[3] = {x, y, z};
    return vrt_patch_plan_box(c, p, p, max_depth, out);
}

int vrt_patch_begin(vrt_ctx *c) {
    if (!c) return VRT_E_INVALID;
    if (!c->have_scene) return vrt_fail(c, VRT_E_STATE, "vrt_patch_begin: no octree uploaded");
    if (c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_patch_begin: a batch is already open");
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
    if (!c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_patch_end: no batch open");
    VRT_HIP(c, hipSetDevice(c->device));
    return patch_device(c);
}

int vrt_patch_apply(vrt_ctx *c, const vrt_patch *patch, const uint32_t *subtree_records, size_t n_records) {
    if (!c || !patch || !subtree_records) return c ? vrt_fail(c, VRT_E_INVALID, "vrt_patch_apply: null argument") : VRT_E_INVALID;
    if (!c->have_scene) return vrt_fail(c, VRT_E_STATE, "vrt_patch_apply: no octree uploaded");
    if (patch->depth < 1 || patch->depth > 15) return vrt_fail(c, VRT_E_INVALID, "vrt_patch_apply: depth out of range");
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
    if (!c->have_scene) return vrt_fail(c, VRT_E_STATE, "vrt_compact: no octree uploaded");
    // the batch's bookkeeping (records_before, rewritten records, repointed cells) indexes the arrays compaction re-lays
    if (c->batch.open) return vrt_fail(c, VRT_E_STATE, "vrt_compact: a patch batch is open (call vrt_patch_end first)");
    VRT_HIP(c, hipSetDevice(c->device));
    VRT_HIP(c, hipDeviceSynchronize());   // dispatches in flight read the old arrays
    vrt::compact_records(c->host_records);
    c->scene_opaque_valid = false;
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

}  // extern "C"
