// vrt_multi.hip -- several GPUs behind the C-ABI of include/vrt.h: peer-visible device memory, stream-ordered flags, and
// vrt_multi (one context per device in one process, frames assembled on the first device). Built into libvrt_hip.so.
//
// Nothing here is a collective: the path has no exchange step while tracing (pixels are independent), and what it does
// afterwards -- bringing the row tiles of one frame together on one device -- is either the trace kernels' own stores
// through xGMI peer mappings (VRT_DELIVER_PEER_STORE) or one pull kernel per image on the assembling device
// (VRT_DELIVER_GATHER). The one-process-per-GPU form with RCCL lives in voxel-raytracer_amd/sharding.py (bench.py).
#include "../../include/vrt.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_multi_create_error;

// rows [0, n_rows) of a compact shard buffer (tile t of the shard = rows [t * tile_rows, ...)) to their place in the frame:
// frame row = shard_row0 + (j / tile_rows) * row_stride + j % tile_rows. words_per_pixel: 1 (rgba8) or 2 (id, dist).
__global__ __launch_bounds__(256) void unshard_kernel(const uint32_t *__restrict__ shard, uint32_t *__restrict__ frame, int width, int n_rows,
                                                      int tile_rows, int row0, int row_stride, int words_per_pixel) {
    const size_t row_words = (size_t)width * (size_t)words_per_pixel;
    const size_t total = (size_t)n_rows * row_words;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t j = i / row_words, x = i - j * row_words;
        const size_t y = (size_t)row0 + (j / (size_t)tile_rows) * (size_t)row_stride + j % (size_t)tile_rows;
        frame[y * row_words + x] = shard[i];
    }
}

}  // namespace

struct vrt_multi {
    std::vector<int> devices;
    std::vector<vrt_ctx *> ctx;
    std::vector<hipEvent_t> traced;     // per device: its share of the current frame has been enqueued / finished
    std::vector<void *> shard_rgba, shard_id;   // VRT_DELIVER_GATHER: per-device compact buffers
    std::vector<size_t> shard_pixels;
    std::vector<void *> band_rgba, band_id, band_shown;   // vrt_multi_dispatch_frame: per-device band + halo (traced, ids, displayed)
    std::vector<size_t> band_pixels;
    hipEvent_t frame_free = nullptr;    // device 0: the consumers of the previous frame have been enqueued before this point
    std::string err;
};

namespace {

int mfail(vrt_multi *m, int code, const std::string &msg) {
    if (m) m->err = msg;
    return code;
}

#define VRTM_HIP(m, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return mfail((m), VRT_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define VRTC_HIP(call)                                  \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return VRT_E_HIP;         \
    } while (0)

}  // namespace

extern "C" {

int vrt_device_alloc(vrt_ctx *c, size_t bytes, void **d_ptr) {
    if (!c || !d_ptr || bytes == 0) return VRT_E_INVALID;
    *d_ptr = nullptr;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    void *p = nullptr;
    VRTC_HIP(hipMalloc(&p, bytes));
    // zero-filled BEFORE the pointer is handed out: the memory is exported over IPC and written from non-blocking streams
    // (flags, frame slots), which have no implicit ordering with a null-stream memset still in flight
    hipStream_t s = (hipStream_t)vrt_stream(c);
    if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        (void)hipFree(p);
        return VRT_E_HIP;
    }
    *d_ptr = p;
    return VRT_OK;
}

int vrt_device_free(vrt_ctx *c, void *d_ptr) {
    if (!c) return VRT_E_INVALID;
    if (!d_ptr) return VRT_OK;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    VRTC_HIP(hipFree(d_ptr));
    return VRT_OK;
}

int vrt_device_read(vrt_ctx *c, const void *d_ptr, void *host, size_t bytes, void *stream) {
    if (!c || !d_ptr || !host) return VRT_E_INVALID;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t)vrt_stream(c);
    VRTC_HIP(hipMemcpyAsync(host, d_ptr, bytes, hipMemcpyDeviceToHost, s));
    VRTC_HIP(hipStreamSynchronize(s));
    return VRT_OK;
}

int vrt_device_write(vrt_ctx *c, void *d_ptr, const void *host, size_t bytes, void *stream) {
    if (!c || !d_ptr || !host) return VRT_E_INVALID;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t)vrt_stream(c);
    VRTC_HIP(hipMemcpyAsync(d_ptr, host, bytes, hipMemcpyHostToDevice, s));
    VRTC_HIP(hipStreamSynchronize(s));
    return VRT_OK;
}

int vrt_device_copy(vrt_ctx *c, void *d_dst, const void *d_src, size_t bytes, void *stream) {
    if (!c || !d_dst || !d_src) return VRT_E_INVALID;
    if (bytes == 0) return VRT_OK;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    hipStream_t s = stream ? (hipStream_t)stream : (hipStream_t)vrt_stream(c);
    VRTC_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDefault, s));   // either side may be another device's memory (peer / IPC mapping)
    return VRT_OK;
}

static_assert(sizeof(hipIpcMemHandle_t) <= VRT_IPC_HANDLE_BYTES, "the handle travels as 64 opaque bytes");

int vrt_ipc_export(vrt_ctx *c, void *d_ptr, uint8_t handle[VRT_IPC_HANDLE_BYTES]) {
    if (!c || !d_ptr || !handle) return VRT_E_INVALID;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    hipIpcMemHandle_t h;
    VRTC_HIP(hipIpcGetMemHandle(&h, d_ptr));
    std::memset(handle, 0, VRT_IPC_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof h);
    return VRT_OK;
}

int vrt_ipc_open(vrt_ctx *c, const uint8_t handle[VRT_IPC_HANDLE_BYTES], void **d_ptr) {
    if (!c || !handle || !d_ptr) return VRT_E_INVALID;
    *d_ptr = nullptr;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    VRTC_HIP(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return VRT_OK;
}

int vrt_ipc_close(vrt_ctx *c, void *d_ptr) {
    if (!c) return VRT_E_INVALID;
    if (!d_ptr) return VRT_OK;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    VRTC_HIP(hipIpcCloseMemHandle(d_ptr));
    return VRT_OK;
}

// The runtime's own stream memory operations (a write / a wait packet in the queue): no kernel of ours spins on a flag,
// so a producer that never arrives stalls a stream, not a compute unit.
int vrt_stream_write_flag(vrt_ctx *c, void *d_flag, uint32_t value, void *stream) {
    if (!c || !d_flag) return VRT_E_INVALID;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    VRTC_HIP(hipStreamWriteValue32(stream ? (hipStream_t)stream : (hipStream_t)vrt_stream(c), d_flag, value, 0));
    return VRT_OK;
}

int vrt_stream_wait_flag(vrt_ctx *c, void *d_flag, uint32_t value, void *stream) {
    if (!c || !d_flag) return VRT_E_INVALID;
    VRTC_HIP(hipSetDevice(vrt_device(c)));
    VRTC_HIP(hipStreamWaitValue32(stream ? (hipStream_t)stream : (hipStream_t)vrt_stream(c), d_flag, value, hipStreamWaitValueGte, 0xffffffffu));
    return VRT_OK;
}

const char *vrt_multi_last_error(const vrt_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }
int vrt_multi_devices(const vrt_multi *m) { return m ? (int)m->devices.size() : VRT_E_INVALID; }
vrt_ctx *vrt_multi_context(vrt_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }
void *vrt_multi_stream(vrt_multi *m) { return (m && !m->ctx.empty()) ? vrt_stream(m->ctx[0]) : nullptr; }

void vrt_destroy_multi(vrt_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (!m->ctx[i]) continue;
        (void)hipSetDevice(m->devices[i]);
        (void)hipStreamSynchronize((hipStream_t)vrt_stream(m->ctx[i]));
        if (i < m->shard_rgba.size()) { (void)hipFree(m->shard_rgba[i]); (void)hipFree(m->shard_id[i]); }
        if (i < m->band_rgba.size()) { (void)hipFree(m->band_rgba[i]); (void)hipFree(m->band_id[i]); (void)hipFree(m->band_shown[i]); }
        if (i < m->traced.size() && m->traced[i]) (void)hipEventDestroy(m->traced[i]);
    }
    if (m->frame_free) { (void)hipSetDevice(m->devices[0]); (void)hipEventDestroy(m->frame_free); }
    for (vrt_ctx *c : m->ctx) vrt_destroy(c);
    delete m;
}

int vrt_create_multi(int n_devices, const int *device_ids, vrt_multi **out) {
    if (!out) return VRT_E_INVALID;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64 || !device_ids) {
        g_multi_create_error = "vrt_create_multi: 1 to 64 devices";
        return VRT_E_INVALID;
    }
    vrt_multi *m = new (std::nothrow) vrt_multi();
    if (!m) return VRT_E_INVALID;
    m->devices.assign(device_ids, device_ids + n_devices);
    m->ctx.assign((size_t)n_devices, nullptr);
    m->traced.assign((size_t)n_devices, nullptr);
    m->shard_rgba.assign((size_t)n_devices, nullptr);
    m->shard_id.assign((size_t)n_devices, nullptr);
    m->shard_pixels.assign((size_t)n_devices, 0);
    for (int i = 0; i < n_devices; ++i) {
        const int r = vrt_create(device_ids[i], &m->ctx[(size_t)i]);
        if (r != VRT_OK) {
            g_multi_create_error = std::string("vrt_create_multi: device ") + std::to_string(device_ids[i]) + ": " + vrt_last_error(nullptr);
            vrt_destroy_multi(m);
            return r;
        }
        hipError_t e = hipSetDevice(device_ids[i]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->traced[(size_t)i], hipEventDisableTiming);
        // every device must be able to store into (and device 0 to read from) the others' memory; the same device listed
        // twice (a rehearsal on one GPU) needs nothing
        if (e == hipSuccess && i > 0 && device_ids[i] != device_ids[0]) {
            int can = 0;
            e = hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[0]);
            if (e == hipSuccess && !can) {
                g_multi_create_error = "vrt_create_multi: device " + std::to_string(device_ids[i]) + " cannot access device " + std::to_string(device_ids[0]);
                vrt_destroy_multi(m);
                return VRT_E_NO_DEVICE;
            }
            if (e == hipSuccess) {
                e = hipDeviceEnablePeerAccess(device_ids[0], 0);   // current device = device_ids[i]
                if (e == hipErrorPeerAccessAlreadyEnabled) { e = hipSuccess; (void)hipGetLastError(); }
            }
            if (e == hipSuccess) {
                e = hipSetDevice(device_ids[0]);
                if (e == hipSuccess) e = hipDeviceEnablePeerAccess(device_ids[i], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { e = hipSuccess; (void)hipGetLastError(); }
            }
        }
        if (e != hipSuccess) {
            g_multi_create_error = std::string("vrt_create_multi: ") + hipGetErrorString(e);
            vrt_destroy_multi(m);
            return VRT_E_HIP;
        }
    }
    if (hipSetDevice(device_ids[0]) != hipSuccess || hipEventCreateWithFlags(&m->frame_free, hipEventDisableTiming) != hipSuccess) {
        g_multi_create_error = "vrt_create_multi: event creation failed";
        vrt_destroy_multi(m);
        return VRT_E_HIP;
    }
    *out = m;
    return VRT_OK;
}

int vrt_multi_upload_octree(vrt_multi *m, const uint8_t *texels, size_t used_bytes, uint32_t tex_dim) {
    if (!m) return VRT_E_INVALID;
    for (vrt_ctx *c : m->ctx) {
        const int r = vrt_upload_octree(c, texels, used_bytes, tex_dim);
        if (r) return mfail(m, r, vrt_last_error(c));
    }
    return VRT_OK;
}

int vrt_multi_set_camera(vrt_multi *m, const float inv_projection[16], const float inv_view[16], const float camera_pos[4]) {
    if (!m) return VRT_E_INVALID;
    for (vrt_ctx *c : m->ctx) {
        const int r = vrt_set_camera(c, inv_projection, inv_view, camera_pos);
        if (r) return mfail(m, r, vrt_last_error(c));
    }
    return VRT_OK;
}

int vrt_multi_set_params(vrt_multi *m, const vrt_params *p) {
    if (!m) return VRT_E_INVALID;
    for (vrt_ctx *c : m->ctx) {
        const int r = vrt_set_params(c, p);
        if (r) return mfail(m, r, vrt_last_error(c));
    }
    return VRT_OK;
}

int vrt_multi_frame_alloc(vrt_multi *m, int width, int height, void **d_rgba8, void **d_id_dist) {
    if (!m || !d_rgba8 || !d_id_dist || width < 1 || height < 1) return VRT_E_INVALID;
    const size_t px = (size_t)width * (size_t)height;
    int r = vrt_device_alloc(m->ctx[0], px * 4, d_rgba8);
    if (r == VRT_OK) r = vrt_device_alloc(m->ctx[0], px * 8, d_id_dist);
    if (r != VRT_OK) {
        (void)vrt_device_free(m->ctx[0], *d_rgba8);
        *d_rgba8 = *d_id_dist = nullptr;
        return mfail(m, r, "vrt_multi_frame_alloc: hipMalloc failed");
    }
    return VRT_OK;
}

int vrt_multi_frame_free(vrt_multi *m, void *d_rgba8, void *d_id_dist) {
    if (!m) return VRT_E_INVALID;
    VRTM_HIP(m, hipSetDevice(m->devices[0]));
    VRTM_HIP(m, hipStreamSynchronize((hipStream_t)vrt_stream(m->ctx[0])));
    (void)vrt_device_free(m->ctx[0], d_rgba8);
    (void)vrt_device_free(m->ctx[0], d_id_dist);
    return VRT_OK;
}

int vrt_multi_dispatch(vrt_multi *m, int width, int height, int tile_rows, int mode, int delivery, void *d_rgba8, void *d_id_dist) {
    if (!m) return VRT_E_INVALID;
    if (width < 1 || height < 1 || tile_rows < 1) return mfail(m, VRT_E_INVALID, "vrt_multi_dispatch: bad frame shape");
    if (delivery != VRT_DELIVER_PEER_STORE && delivery != VRT_DELIVER_GATHER) return mfail(m, VRT_E_INVALID, "vrt_multi_dispatch: unknown delivery");
    const int n = (int)m->ctx.size();
    hipStream_t s0 = (hipStream_t)vrt_stream(m->ctx[0]);
    // the other devices may not touch the frame before what device 0's stream already holds (the consumers of the previous
    // frame in these buffers) has run
    VRTM_HIP(m, hipSetDevice(m->devices[0]));
    VRTM_HIP(m, hipEventRecord(m->frame_free, s0));
    for (int i = 0; i < n; ++i) {
        vrt_ctx *c = m->ctx[(size_t)i];
        hipStream_t s = (hipStream_t)vrt_stream(c);
        VRTM_HIP(m, hipSetDevice(m->devices[(size_t)i]));
        if (i > 0) VRTM_HIP(m, hipStreamWaitEvent(s, m->frame_free, 0));
        int r;
        if (delivery == VRT_DELIVER_PEER_STORE) {
            r = vrt_dispatch_tiles(c, width, height, tile_rows, i, n, mode, d_rgba8, d_id_dist, nullptr);
        } else {
            const int rows = vrt_shard_rows(height, tile_rows, i, n);
            const size_t px = (size_t)(rows > 0 ? rows : 0) * (size_t)width;
            if (px > m->shard_pixels[(size_t)i]) {
                VRTM_HIP(m, hipStreamSynchronize(s));
                (void)hipFree(m->shard_rgba[(size_t)i]);
                (void)hipFree(m->shard_id[(size_t)i]);
                m->shard_rgba[(size_t)i] = m->shard_id[(size_t)i] = nullptr;
                m->shard_pixels[(size_t)i] = 0;
                VRTM_HIP(m, hipMalloc(&m->shard_rgba[(size_t)i], px * 4));
                VRTM_HIP(m, hipMalloc(&m->shard_id[(size_t)i], px * 8));
                m->shard_pixels[(size_t)i] = px;
            }
            r = rows > 0 ? vrt_dispatch_shard(c, width, height, tile_rows, i, n, mode, d_rgba8 ? m->shard_rgba[(size_t)i] : nullptr,
                                              d_id_dist ? m->shard_id[(size_t)i] : nullptr, nullptr)
                         : VRT_OK;
        }
        if (r) return mfail(m, r, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + vrt_last_error(c));
        VRTM_HIP(m, hipEventRecord(m->traced[(size_t)i], s));
    }
    // device 0's stream is where the finished frame is: it waits for every device's share
    VRTM_HIP(m, hipSetDevice(m->devices[0]));
    for (int i = 1; i < n; ++i) VRTM_HIP(m, hipStreamWaitEvent(s0, m->traced[(size_t)i], 0));
    if (delivery == VRT_DELIVER_GATHER) {
        for (int i = 0; i < n; ++i) {
            const int rows = vrt_shard_rows(height, tile_rows, i, n);
            if (rows <= 0) continue;
            const size_t px = (size_t)rows * (size_t)width;
            const unsigned blocks = (unsigned)((px + 255) / 256 < 4096 ? (px + 255) / 256 : 4096);
            if (d_rgba8)
                hipLaunchKernelGGL(unshard_kernel, dim3(blocks), dim3(256), 0, s0, (const uint32_t *)m->shard_rgba[(size_t)i], (uint32_t *)d_rgba8,
                                   width, rows, tile_rows, i * tile_rows, tile_rows * n, 1);
            if (d_id_dist)
                hipLaunchKernelGGL(unshard_kernel, dim3(blocks), dim3(256), 0, s0, (const uint32_t *)m->shard_id[(size_t)i], (uint32_t *)d_id_dist,
                                   width, rows, tile_rows, i * tile_rows, tile_rows * n, 2);
            VRTM_HIP(m, hipGetLastError());
        }
        // the shard buffers are free again once the pull kernels have run: the next frame's traces wait for that
        // through frame_free, which is recorded on s0 at the start of the next call
    }
    return VRT_OK;
}

// The frame the reference SHOWS (dispatch, then the display pass) over the devices of m: row bands with a 20-row halo, each
// device traces and filters its band, the band's displayed rows are copied into d_shown_rgba8 on device_ids[0].
int vrt_multi_dispatch_frame(vrt_multi *m, int width, int height, int mode, void *d_shown_rgba8) {
    if (!m || !d_shown_rgba8) return m ? mfail(m, VRT_E_INVALID, "vrt_multi_dispatch_frame: null frame") : VRT_E_INVALID;
    if (width < 1 || height < 1) return mfail(m, VRT_E_INVALID, "vrt_multi_dispatch_frame: bad frame shape");
    constexpr int kHalo = 20;   // quad.frag's largest radius
    const int n = (int)m->ctx.size();
    if (m->band_pixels.size() != (size_t)n) {
        m->band_rgba.assign((size_t)n, nullptr); m->band_id.assign((size_t)n, nullptr); m->band_shown.assign((size_t)n, nullptr);
        m->band_pixels.assign((size_t)n, 0);
    }
    hipStream_t s0 = (hipStream_t)vrt_stream(m->ctx[0]);
    VRTM_HIP(m, hipSetDevice(m->devices[0]));
    VRTM_HIP(m, hipEventRecord(m->frame_free, s0));
    const int tiles = (height + 7) / 8;
    for (int i = 0; i < n; ++i) {
        vrt_ctx *c = m->ctx[(size_t)i];
        hipStream_t s = (hipStream_t)vrt_stream(c);
        VRTM_HIP(m, hipSetDevice(m->devices[(size_t)i]));
        if (i > 0) VRTM_HIP(m, hipStreamWaitEvent(s, m->frame_free, 0));
        int b0 = tiles * i / n * 8, b1 = tiles * (i + 1) / n * 8;
        if (b0 > height) b0 = height;
        if (b1 > height) b1 = height;
        if (b1 > b0) {
            const int h0 = b0 - kHalo > 0 ? b0 - kHalo : 0, h1 = b1 + kHalo < height ? b1 + kHalo : height;
            const size_t px = (size_t)(h1 - h0) * (size_t)width;
            if (px > m->band_pixels[(size_t)i]) {
                VRTM_HIP(m, hipStreamSynchronize(s));
                (void)hipFree(m->band_rgba[(size_t)i]); (void)hipFree(m->band_id[(size_t)i]); (void)hipFree(m->band_shown[(size_t)i]);
                m->band_rgba[(size_t)i] = m->band_id[(size_t)i] = m->band_shown[(size_t)i] = nullptr;
                m->band_pixels[(size_t)i] = 0;
                VRTM_HIP(m, hipMalloc(&m->band_rgba[(size_t)i], px * 4));
                VRTM_HIP(m, hipMalloc(&m->band_id[(size_t)i], px * 8));
                VRTM_HIP(m, hipMalloc(&m->band_shown[(size_t)i], px * 4));
                m->band_pixels[(size_t)i] = px;
            }
            // vrt_dispatch_rows addresses its images by frame row: the base is shifted up by h0 rows
            char *rg = (char *)m->band_rgba[(size_t)i], *id = (char *)m->band_id[(size_t)i], *sh = (char *)m->band_shown[(size_t)i];
            int r = vrt_dispatch_rows(c, width, height, h0, h1, mode, rg - (size_t)h0 * (size_t)width * 4, id - (size_t)h0 * (size_t)width * 8, nullptr);
            if (!r) r = vrt_denoise(c, width, h1 - h0, rg, id, sh, nullptr);
            if (r) return mfail(m, r, std::string("device ") + std::to_string(m->devices[(size_t)i]) + ": " + vrt_last_error(c));
            VRTM_HIP(m, hipMemcpyAsync((char *)d_shown_rgba8 + (size_t)b0 * (size_t)width * 4, sh + (size_t)(b0 - h0) * (size_t)width * 4,
                                       (size_t)(b1 - b0) * (size_t)width * 4, hipMemcpyDefault, s));
        }
        VRTM_HIP(m, hipEventRecord(m->traced[(size_t)i], s));
    }
    VRTM_HIP(m, hipSetDevice(m->devices[0]));
    for (int i = 1; i < n; ++i) VRTM_HIP(m, hipStreamWaitEvent(s0, m->traced[(size_t)i], 0));
    return VRT_OK;
}

int vrt_multi_synchronize(vrt_multi *m) {
    if (!m) return VRT_E_INVALID;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        VRTM_HIP(m, hipSetDevice(m->devices[i]));
        VRTM_HIP(m, hipStreamSynchronize((hipStream_t)vrt_stream(m->ctx[i])));
    }
    return VRT_OK;
}

}  // extern "C"
