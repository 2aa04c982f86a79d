/*
 * vrt.h -- C-ABI of the MI355X (gfx950) ray-casting layer: libvrt_hip.so.
 *
 * Drop-in boundary for the compute-shader dispatch of pedroand6/Voxel-Raytracer.
 * Each entry point replaces one piece of the reference's GL plumbing around
 * `glDispatchCompute` (reference paths are relative to the upstream repo root):
 *
 *   vrt_create / vrt_destroy   GL object setup/teardown        src/main.cpp:432-474, 973-983
 *   vrt_upload_octree          updateGPUTexture()/glTexImage3D src/main.cpp:264-311
 *   vrt_set_camera             Camera UBO glBufferSubData      src/main.cpp:643-656, 807-813, 916-917
 *   vrt_set_params             the seven glUniform* calls      src/main.cpp:689-695, 932-938
 *   vrt_dispatch*              glMemoryBarrier+glDispatchCompute src/main.cpp:941-946
 *                              (shader: shaders/raytracing.comp:624-645)
 *
 * Plain pointers and sizes only; no C++/torch types cross this boundary and no
 * exception escapes. Every call returns 0 on success or a negative VRT_E_*;
 * vrt_last_error() gives the text. One context per host thread (thread-
 * compatible, not thread-safe). There is no CPU fallback: without a HIP device
 * vrt_create fails with VRT_E_NO_DEVICE.
 */
#ifndef VRT_H
#define VRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_OK 0
#define VRT_E_INVALID (-1)    /* bad argument */
#define VRT_E_NO_DEVICE (-2)  /* no usable HIP device / HIP runtime error at create */
#define VRT_E_HIP (-3)        /* HIP runtime call failed */
#define VRT_E_MALFORMED (-4)  /* texel stream cannot be a flattened octree */
#define VRT_E_STATE (-5)      /* call order (e.g. dispatch before upload) */

/* which subset of pathTrace (raytracing.comp:435-622) a dispatch evaluates */
#define VRT_MODE_PRIMARY 0         /* primary ray, direct term unshadowed */
#define VRT_MODE_PRIMARY_SHADOW 1  /* + notInShadow() ray per opaque hit (comp:333-377, 587) */
#define VRT_MODE_FULL 2            /* the whole shader: glass stack, diffuse bounce, RNG */

typedef struct vrt_ctx vrt_ctx;

/* The shader's scalar uniforms (raytracing.comp:27-39). u_texDim travels with
 * vrt_upload_octree. Defaults = the values src/main.cpp:478-483,638 sets. */
typedef struct vrt_params {
    float voxel_scale;          /* u_voxelScale            (1.0)              */
    int32_t world_min[3];       /* u_worldBoundsMin        (-1023,-1023,-1023) */
    int32_t world_max[3];       /* u_worldBoundsMax        (1024,1024,1024)    */
    float global_light[4];      /* globalLight             (1,1,1,1)           */
    float light_dir[3];         /* lightDir  normalize(.3481553,.870388,.3481553) */
    int32_t highlighted[3];     /* u_highlightedVoxel      (-1,-1,-1)          */
} vrt_params;

typedef struct vrt_scene_info {
    uint32_t tex_dim;           /* u_texDim given at upload */
    uint32_t n_texels;          /* texels in the uploaded stream */
    uint32_t n_records;         /* 8-byte device records (internal + leaf) */
    uint32_t n_internal;        /* internal nodes */
    uint32_t n_leaves;          /* leaf nodes */
    uint32_t max_depth;         /* deepest node below the root */
    uint32_t lds_records;       /* records of the level-order prefix staged in LDS */
    uint32_t reserved;
} vrt_scene_info;

int vrt_create(int device_id, vrt_ctx **out);
void vrt_destroy(vrt_ctx *ctx);
/* ctx may be NULL: returns the message of the last failed vrt_create on this thread */
const char *vrt_last_error(const vrt_ctx *ctx);

/* Fills *p with the reference defaults. */
void vrt_default_params(vrt_params *p);
int vrt_set_params(vrt_ctx *ctx, const vrt_params *p);

/* texels: the byte stream octree_texture() returns (4 bytes per texel, root
 * header at texel 0), used_bytes its size, tex_dim = ceil(cbrt(texels)).
 * The library copies and re-lays it out for the device; the caller keeps
 * ownership. texels == NULL / used_bytes == 0 uploads an empty world. */
int vrt_upload_octree(vrt_ctx *ctx, const uint8_t *texels, size_t used_bytes, uint32_t tex_dim);
int vrt_get_scene_info(const vrt_ctx *ctx, vrt_scene_info *info);

/* EXTENSION: edits without re-flattening and re-uploading the tree (the reference does both on every build / destroy
 * click, src/main.cpp:903-914). After the host octree has been edited at voxel (x, y, z):
 *   1. vrt_patch_plan() names the deepest ancestor A of the voxel (depth <= max_depth) whose sub-tree can be replaced
 *      on the device: depth below the root and the child index taken at each level;
 *   2. the host library says whether A is still an internal node (vrth_octree_node_state) -- if not, plan again with
 *      max_depth = depth - 1 -- and emits A's new sub-tree (vrth_octree_path_records: just the nodes that contain the
 *      voxel, or vrth_octree_subtree_records: all of it);
 *   3. vrt_patch_apply() appends those records, rewrites A's record, rebuilds A's part of the wide layout and copies
 *      only what changed to the device (after waiting for dispatches in flight).
 * vrt_patch_plan returns VRT_E_STATE when no ancestor qualifies (use vrt_upload_octree / vrt_upload_records).
 * Replaced sub-trees stay allocated until they outweigh the tree, then vrt_patch_plan compacts the arrays (vrt_compact);
 * vrt_get_scene_info().n_records shows both.
 * Pixels after a patch equal those after a full upload of the edited tree. */
typedef struct vrt_patch {
    int32_t depth;       /* of A below the root, >= 1 */
    uint8_t path[16];    /* child index taken at levels 0 .. depth-1 */
} vrt_patch;
int vrt_patch_plan(vrt_ctx *ctx, int x, int y, int z, int max_depth, vrt_patch *out);
/* The same for an edit that touched a whole BOX of voxels [lo, hi] (inclusive) -- a fill, an explosion: the deepest patchable
 * ancestor whose cube holds the box. The host library then emits ONE sub-tree for it (vrth_octree_box_records: only the nodes
 * that meet the box are walked, the rest are "keep" records) and ONE vrt_patch_apply replaces it: a 16^3 fill is one patch,
 * not 4,096. */
int vrt_patch_plan_box(vrt_ctx *ctx, const int32_t lo[3], const int32_t hi[3], int max_depth, vrt_patch *out);
int vrt_patch_apply(vrt_ctx *ctx, const vrt_patch *patch, const uint32_t *subtree_records, size_t n_records);
/* A batch of edits (a brush stroke, an explosion; src/main.cpp:843-914 re-flattens once per click): between
 * vrt_patch_begin and vrt_patch_end, vrt_patch_plan / vrt_patch_apply work on the library's host copy of the structures
 * only -- each plan sees the patches before it -- and vrt_patch_end sends everything the batch appended or rewrote to the
 * device in one go (one wait for dispatches in flight, one upload). Dispatching with a batch open is an error
 * (VRT_E_STATE). A patch the library refuses (VRT_E_STATE / VRT_E_MALFORMED) changes nothing; a device failure in
 * vrt_patch_end drops the scene (upload again). Outside a batch vrt_patch_apply is begin + apply + end. */
int vrt_patch_begin(vrt_ctx *ctx);
int vrt_patch_end(vrt_ctx *ctx);
/* Reclaims what patches left behind (replaced child blocks and wide nodes): the device arrays are re-laid from the live
 * tree, without a texel stream. vrt_patch_plan does this by itself once the garbage outweighs the tree; pixels do not
 * change. */
int vrt_compact(vrt_ctx *ctx);

/* EXTENSION (not a reference interface): upload the device record array itself -- 2 x uint32 per record,
 * level order, root first; internal: {child_mask | leaf_mask << 8, first child index}, leaf:
 * {R | G<<8 | B<<16 | alpha<<24, refr | illum<<8 | k<<16} -- as vrth_world_records() (vrt_host.h) emits it
 * straight from the pointer octree. It replaces the reference's full re-flatten + re-upload on every edit
 * (src/main.cpp:903-914 -> :264-311) and is not limited to 2^23 texels (src/octree.cpp:556-570).
 * tex_dim must still be ceil(cbrt(_octree_texel_size(tree))): it feeds the voxelID output. */
int vrt_upload_records(vrt_ctx *ctx, const uint32_t *records, size_t n_records, uint32_t tex_dim);

/* Column-major mat4 x2 + vec4, exactly the std140 Camera block (comp:17-21). */
int vrt_set_camera(vrt_ctx *ctx, const float inv_projection[16], const float inv_view[16],
                   const float camera_pos[4]);

/* Synchronous whole-frame dispatch into HOST buffers:
 *   out_rgba8   width*height*4 bytes  (image binding 0, rgba8; row 0 = bottom, v = -1)
 *   out_id_dist width*height*2 int32  (image binding 3, rg32i = voxelID, dist)
 * Either pointer may be NULL. Any width/height >= 1 is accepted. */
int vrt_dispatch(vrt_ctx *ctx, int width, int height, int mode, uint8_t *out_rgba8, int32_t *out_id_dist);

/* The same, asynchronous and double-buffered: the call enqueues the trace and the two copies to the host on one of two
 * internal lanes (a stream and a pair of device images each) and returns a ticket (0 or 1); vrt_dispatch_wait(ticket)
 * blocks until that frame's host buffers are complete. While frame i crosses PCIe, frame i+1 is traced: a loop that
 * keeps two frames in flight runs at the copy's rate (12 B/pixel over PCIe: 24.9 MB, >= 0.40 ms at the link's 63 GB/s
 * for a 1080p frame -- no arrangement of copies brings both images of that frame to the host faster) instead of trace +
 * copy. The host buffers should be pinned (vrt_host_alloc): into pageable memory the runtime stages the copy and the
 * call blocks for most of it. At most two frames in flight: a third call waits for the older ticket itself. */
int vrt_dispatch_async(vrt_ctx *ctx, int width, int height, int mode, uint8_t *out_rgba8, int32_t *out_id_dist, int *ticket);
int vrt_dispatch_wait(vrt_ctx *ctx, int ticket);
/* page-locked host memory for those buffers (hipHostMalloc / hipHostFree) */
int vrt_host_alloc(vrt_ctx *ctx, size_t bytes, void **host_ptr);
int vrt_host_free(vrt_ctx *ctx, void *host_ptr);

/* Stream-ordered dispatch into DEVICE buffers laid out as full frames; only
 * rows [row_begin, row_end) are traced and written (row sharding across GPUs).
 * stream: a hipStream_t, or NULL for the context's own stream. Returns after
 * enqueueing. */
int vrt_dispatch_rows(vrt_ctx *ctx, int width, int height, int row_begin, int row_end, int mode,
                      void *d_rgba8, void *d_id_dist, void *stream);

/* Interleaved row-tile sharding: the frame is cut into tiles of tile_rows rows;
 * shard s of n_shards owns tiles t with t % n_shards == s. Output buffers are
 * COMPACT: the shard's tiles back to back in tile order (each W*tile_rows
 * pixels, the last one possibly shorter). */
int vrt_dispatch_shard(vrt_ctx *ctx, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       void *d_rgba8, void *d_id_dist, void *stream);
/* rows (and pixels = rows*width) a shard owns under that scheme */
int vrt_shard_rows(int height, int tile_rows, int shard, int n_shards);
/* The same tiles written at their place in a FULL frame (width*height pixels) instead of a compact buffer: the frame
 * may live on this device, on a peer device whose memory this one can reach (hipDeviceEnablePeerAccess, or an
 * allocation opened with vrt_ipc_open in another process) -- the shards of one frame then land in ONE framebuffer
 * with no gather step: the kernels' own stores cross xGMI. */
int vrt_dispatch_tiles(vrt_ctx *ctx, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       void *d_frame_rgba8, void *d_frame_id_dist, void *stream);

/* Device memory that other processes of the node can map (one process per GPU: the frame lives on one rank, the
 * others store into it). vrt_device_alloc is hipMalloc on the context's device (zero-filled); vrt_ipc_export fills a
 * 64-byte handle another process passes to vrt_ipc_open, which returns the address of the same memory in ITS address
 * space (on its context's device: peer access over xGMI); vrt_ipc_close unmaps it. Needs
 * HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf handles) in both processes. */
#define VRT_IPC_HANDLE_BYTES 64
int vrt_device_alloc(vrt_ctx *ctx, size_t bytes, void **d_ptr);
int vrt_device_free(vrt_ctx *ctx, void *d_ptr);
/* synchronous copies between such memory and the host, after the work enqueued on `stream` (NULL: the context's) */
int vrt_device_read(vrt_ctx *ctx, const void *d_ptr, void *host, size_t bytes, void *stream);
int vrt_device_write(vrt_ctx *ctx, void *d_ptr, const void *host, size_t bytes, void *stream);
/* Device-to-device copy ordered on `stream` (NULL: the context's), asynchronous. Either pointer may be a mapping of another
 * rank's memory (vrt_ipc_open) or another device's (vrt_multi): how a rank hands a finished block of rows -- e.g. its band
 * of the DISPLAYED image after vrt_denoise -- to the rank that shows it. */
int vrt_device_copy(vrt_ctx *ctx, void *d_dst, const void *d_src, size_t bytes, void *stream);
int vrt_ipc_export(vrt_ctx *ctx, void *d_ptr, uint8_t handle[VRT_IPC_HANDLE_BYTES]);
int vrt_ipc_open(vrt_ctx *ctx, const uint8_t handle[VRT_IPC_HANDLE_BYTES], void **d_ptr);
int vrt_ipc_close(vrt_ctx *ctx, void *d_ptr);
/* Stream-ordered flags in device memory (a uint32 per flag): vrt_stream_write_flag makes `stream` store `value`
 * once everything enqueued before it has finished; vrt_stream_wait_flag makes `stream` wait until *d_flag >= value.
 * With the flag in an IPC-mapped allocation this is how a producer rank tells the consumer rank "my tiles of frame i
 * have landed" (and the consumer tells it "buffer k is free again") without the host or a collective in the loop. */
int vrt_stream_write_flag(vrt_ctx *ctx, void *d_flag, uint32_t value, void *stream);
int vrt_stream_wait_flag(vrt_ctx *ctx, void *d_flag, uint32_t value, void *stream);

/* ---- several GPUs behind one handle (reference: one GL context, src/main.cpp:432-474) -------------------------------
 * vrt_create_multi(n, device_ids) makes one context per device in THIS process (no Python, no launcher) and enables
 * peer access from every device to device_ids[0]. Scene, camera and uniforms are replicated by the vrt_multi_* setters.
 * vrt_multi_dispatch traces one frame, the devices sharing it by interleaved tile_rows-row tiles, into full-frame DEVICE
 * buffers on device_ids[0] (vrt_multi_frame_alloc): delivery VRT_DELIVER_PEER_STORE lets every device's kernel store its
 * tiles straight into those buffers (peer mappings over xGMI, no gather); VRT_DELIVER_GATHER traces into per-device
 * compact shard buffers and lets device 0 pull and un-interleave them with one copy kernel per image (reads over xGMI).
 * Returns after enqueueing; vrt_multi_synchronize waits for all devices. The frame is complete on device_ids[0] once
 * the call's work on stream vrt_multi_stream(m) has finished. */
typedef struct vrt_multi vrt_multi;
#define VRT_DELIVER_PEER_STORE 0
#define VRT_DELIVER_GATHER 1
int vrt_create_multi(int n_devices, const int *device_ids, vrt_multi **out);
void vrt_destroy_multi(vrt_multi *m);
const char *vrt_multi_last_error(const vrt_multi *m);   /* m may be NULL: the last failed vrt_create_multi on this thread */
int vrt_multi_devices(const vrt_multi *m);
vrt_ctx *vrt_multi_context(vrt_multi *m, int i);        /* the i-th device's context (for per-device calls) */
int vrt_multi_upload_octree(vrt_multi *m, const uint8_t *texels, size_t used_bytes, uint32_t tex_dim);
int vrt_multi_set_camera(vrt_multi *m, const float inv_projection[16], const float inv_view[16], const float camera_pos[4]);
int vrt_multi_set_params(vrt_multi *m, const vrt_params *p);
int vrt_multi_frame_alloc(vrt_multi *m, int width, int height, void **d_rgba8, void **d_id_dist);   /* on device_ids[0] */
int vrt_multi_frame_free(vrt_multi *m, void *d_rgba8, void *d_id_dist);
int vrt_multi_dispatch(vrt_multi *m, int width, int height, int tile_rows, int mode, int delivery, void *d_rgba8,
                       void *d_id_dist);
/* The frame the reference SHOWS -- dispatch (src/main.cpp:946) then the display pass (:951-967, shaders/quad.frag) -- over the
 * devices of m: the frame is cut into one band of rows per device (multiples of 8), every device traces its band plus a
 * 20-row halo on either side (the display pass reads up to 20 rows away), filters that sub-image and copies the band's
 * displayed rows into d_shown_rgba8 (W*H packed rgba8 on device_ids[0], e.g. from vrt_multi_frame_alloc). Only the displayed
 * image crosses devices. Returns after enqueueing; the frame is complete once vrt_multi_stream(m)'s work has finished. */
int vrt_multi_dispatch_frame(vrt_multi *m, int width, int height, int mode, void *d_shown_rgba8);
int vrt_multi_synchronize(vrt_multi *m);
void *vrt_multi_stream(vrt_multi *m);                    /* device_ids[0]'s stream: consumers of the frame order themselves after it */

/* EXTENSION: up to 4 views of the uploaded scene in ONE launch -- frames that are known together (a stereo pair,
 * the next frames of a camera path, the views of a rig). Each view brings its own camera block (what
 * vrt_set_camera takes) and its own compact shard buffers (what vrt_dispatch_shard takes); scene, uniforms and the
 * row sharding are shared. Pixels are those of n_views separate vrt_dispatch_shard calls; the launch is shorter
 * than their sum because one view's last waves no longer drain an otherwise idle chip. DEVICE pointers,
 * stream-ordered, returns after enqueueing. */
typedef struct vrt_view {
    float inv_projection[16];
    float inv_view[16];
    float camera_pos[4];
    void *d_rgba8;      /* rows_of_shard * width packed rgba8, or NULL */
    void *d_id_dist;    /* rows_of_shard * width int2 (voxelID, dist), or NULL */
} vrt_view;
int vrt_dispatch_views(vrt_ctx *ctx, int width, int height, int tile_rows, int shard, int n_shards, int mode,
                       const vrt_view *views, int n_views, void *stream);

/* Repeats vrt_dispatch_rows `iters` times on `stream` with a hipEvent pair
 * around every launch and returns each launch's duration (ms) in ms_out[iters].
 * Blocks until done. */
int vrt_dispatch_timed(vrt_ctx *ctx, int width, int height, int row_begin, int row_end, int mode,
                       void *d_rgba8, void *d_id_dist, void *stream, int iters, float *ms_out);

/* The display pass that consumes the two images in the reference's frame loop (the fullscreen
 * quad drawn by src/main.cpp:951-967 with shaders/quad.frag:22-83): ID-aware box blur, radius
 * clamp(int(200/sqrt(max(1,dist))), 1, 20), only pixels with the centre's voxelID contribute.
 * DEVICE pointers, full frames (W*H packed rgba8 / W*H int2), stream-ordered; out must not alias in. */
int vrt_denoise(vrt_ctx *ctx, int width, int height, const void *d_rgba8, const void *d_id_dist, void *d_out_rgba8,
                void *stream);
/* the same through HOST buffers, synchronous */
int vrt_denoise_host(vrt_ctx *ctx, int width, int height, const uint8_t *rgba8, const int32_t *id_dist,
                     uint8_t *out_rgba8);
/* EXTENSION: one whole frame of the reference's loop -- dispatch (src/main.cpp:946) then the display pass
 * (:951-967) -- with both intermediate images kept on the device; only what the caller asks for comes back.
 * HOST pointers, synchronous. out_shown_rgba8 receives what the reference puts on screen; out_rgba8 and
 * out_id_dist (either may be NULL) the two images the dispatch wrote. */
int vrt_dispatch_frame(vrt_ctx *ctx, int width, int height, int mode, uint8_t *out_shown_rgba8, uint8_t *out_rgba8,
                       int32_t *out_id_dist);

/* Per-launch timing of the dispatches that follow: a hipEvent pair is attached
 * to the kernel's dispatch packet (hipExtLaunchKernel), so it reads the kernel's
 * own begin-to-end time on the stream it runs on, for up to max_launches
 * launches (0 switches it off). vrt_profile_read waits for the recorded
 * launches, writes their durations (ms) and returns how many. */
int vrt_set_profiling(vrt_ctx *ctx, int max_launches);
/* time only every `every`-th launch (default 1): timing every launch costs a few percent of the frame rate */
int vrt_set_profiling_stride(vrt_ctx *ctx, int every);
int vrt_profile_read(vrt_ctx *ctx, float *ms_out, int cap);

int vrt_synchronize(vrt_ctx *ctx);
/* the context's own stream (hipStream_t) and device ordinal */
void *vrt_stream(vrt_ctx *ctx);
int vrt_device(const vrt_ctx *ctx);

/* kernel variant selection (0 = default). The shipped library holds the default and the fallbacks its dispatcher may
 * take (variants 0, 1, 4, 20, 22); the A/B variants exist only in a `make AB=1` build: vrt_variant_available() says
 * which, vrt_set_variant() returns VRT_E_INVALID for the others. */
int vrt_set_variant(vrt_ctx *ctx, int variant);
int vrt_variant_available(int variant);

/* Feedback scheduling of the tracing kernel and the display pass (on by default, period 16). Frames that repeat a
 * launch shape on a stream -- the reference's loop dispatches the same W x H every frame (main.cpp:946) -- start
 * their tiles heaviest first, using the tile times measured on the shape's second launch (the first may be cold) and
 * on every `period`-th one after it (period 1: on all of them); this shortens the tail of a launch (1080p dragon:
 * primary rays -10 %, full shader -13 %, display pass -17 %); a jump of the camera triggers a fresh measurement at
 * once. Pixels do not depend on it. period = 0 switches it off: tiles then start in row-major order. Applies to
 * one-view launches of the default variant in all three modes and to vrt_denoise / vrt_dispatch_frame. */
int vrt_set_tile_scheduling(vrt_ctx *ctx, int period);

/* Switches of the dispatcher. Pixels never depend on them: both settings of each are held to the same oracle frames by the parity
 * suite, which is what they exist for (and for A/B timing).
 *   VRT_OPT_RAY_TABLES     1 (default): views whose inverse projection has the shape of a perspective or orthographic matrix read the
 *                          per-column / per-row part of ray generation (raytracing.comp:626-634) from tables made once per projection;
 *                          0: every launch runs the shader's own prologue.
 *   VRT_OPT_EMPTY_OCTANTS  1 (default): when the tree is empty outside one aligned cube (every scene loaded at the origin of the
 *                          reference's [-1023,1024)^3 world), rays that leave it end there and the deepest node that still holds
 *                          everything stands in for the root; 2: the same without the tighter root; 0: off (rays walk the empty octants).
 *   VRT_OPT_DISPLAY_KERNEL 0 (default): the display pass sums two pixels per lane and every wave takes the cheaper of two walks over its
 *                          staged window: the rows and column segments any of its 64 lanes needs, the same for all lanes (faces that fill
 *                          the window: close-ups), or every pixel the box its own voxel face occupies (faces small against the window:
 *                          1080p dragon frame 0.186 -> 0.123 ms with the staging changes that came with it); 2 / 3: always the first / the second walk (A/B, tests); 1: one pixel per
 *                          lane (round 1's kernel; exists in `make AB=1` builds only, VRT_E_INVALID otherwise).
 *   VRT_OPT_FULL_OPAQUE    VRT_MODE_FULL where pathTrace cannot branch: in a scene without translucent voxels seen from empty space it is
 *                          the primary ray, a shadow ray and ONE diffuse bounce ray that spawns nothing (comp:573-616), so the 8-deep ray
 *                          stack is never used. 6 (default): such launches run a kernel that holds no stack -- primary + shadow stage, then
 *                          the bounce stage in the same wave, 80 registers instead of 96 + 560 B of scratch: 0.137 against 0.172 ms on the
 *                          1080p dragon frame; 5, 7: the same built for five / seven waves per SIMD; 1: the two stages as two kernels with
 *                          a 20-byte seed per pixel between them (0.158 ms: the experiment the one-kernel form came from); 0: always the
 *                          general kernel. Scenes with a translucent voxel and eyes inside a medium take the general kernel whatever the
 *                          setting.
 *   VRT_OPT_HEAVY_TILES    1 (default): the general VRT_MODE_FULL kernel, on launches the feedback scheduler has an order for, traces the
 *                          few heaviest groups of tiles (those above 3/4 of the heaviest one's time, at most 64, when the heaviest tile outlasts 3/4 of its even share of the frame) as eight
 *                          waves per 8x8 tile instead of one. A frame of a translucent scene is as long as its longest wave -- dozens of
 *                          rays one after the other, each round as long as the longest of the wave's marches; with 8 pixels per wave that
 *                          is the longest of 8 instead of 64 (profiles/r03_room_critical_path.txt). 0: every tile is one wave. */
#define VRT_OPT_RAY_TABLES 1
#define VRT_OPT_EMPTY_OCTANTS 2
#define VRT_OPT_DISPLAY_KERNEL 3
#define VRT_OPT_FULL_OPAQUE 4
#define VRT_OPT_HEAVY_TILES 5
int vrt_set_option(vrt_ctx *ctx, int option, int value);

/* The feedback scheduler's order, read and overridden. vrt_get_tile_order copies the current workgroup-group order for the shape last
 * launched on `stream` (NULL: the context's) into out[cap] and returns the number of groups n (0 while no order has been derived);
 * when cap > n, out[n] receives how many groups at the head of the order VRT_MODE_FULL launches trace as part-tile waves
 * (VRT_OPT_HEAVY_TILES; 0 for shapes of other modes).
 * vrt_set_tile_order(enable = 1) makes one-view launches of the default kernel take caller-owned DEVICE buffers instead of the
 * scheduler's: d_group_order (a permutation of the launch's groups of four 8x8 tiles, or NULL: row-major) and d_tile_cost (one uint32
 * per tile, receives each tile's clock ticks, or NULL); enable = 0 hands the launches back to the scheduler. Any permutation renders
 * the same pixels. */
long vrt_get_tile_order(vrt_ctx *ctx, void *stream, uint32_t *out, size_t cap);
int vrt_set_tile_order(vrt_ctx *ctx, int enable, const void *d_group_order, void *d_tile_cost);

const char *vrt_version(void);

#ifdef __cplusplus
}
#endif
#endif
