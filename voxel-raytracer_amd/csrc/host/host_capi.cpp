// host_capi.cpp -- C-ABI of include/vrt_host.h over the C++ host library.
#include "../../../include/vrt_host.h"

#include <Camera.hpp>
#include <octree.hpp>
#include <voxReader.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

bool vrt_load_vox_memory(const uint8_t *data, size_t len, Octree *tree, int offsetX, int offsetY, int offsetZ,
                         long *inserted);

struct vrth_world {
    Octree *root;
};

namespace {
inline IVector3 iv3(int x, int y, int z) {
    IVector3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}
inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

void put_u32(std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
void put_tag(std::vector<uint8_t> &b, const char *t) { b.insert(b.end(), t, t + 4); }

}  // namespace

extern "C" {

const char *vrth_version(void) { return "vrt-host 0.1"; }

vrth_world *vrth_world_create(const int32_t *mn, const int32_t *mx) {
    vrth_world *w = new (std::nothrow) vrth_world();
    if (!w) return NULL;
    const IVector3 lo = mn ? iv3(mn[0], mn[1], mn[2]) : iv3(-1023, -1023, -1023);
    const IVector3 hi = mx ? iv3(mx[0], mx[1], mx[2]) : iv3(1024, 1024, 1024);
    w->root = octree_create(NULL, lo, hi);
    if (!w->root) { delete w; return NULL; }
    return w;
}

void vrth_world_destroy(vrth_world *w) {
    if (!w) return;
    octree_delete(w->root);
    delete w;
}

void *vrth_world_root(vrth_world *w) { return w ? w->root : NULL; }

int vrth_world_load_vox(vrth_world *w, const char *path, int ox, int oy, int oz) {
    if (!w) return -1;
    return load_vox_file(path, w->root, ox, oy, oz) ? 1 : 0;
}

int vrth_world_load_vox_mem(vrth_world *w, const uint8_t *data, size_t len, int ox, int oy, int oz, long *inserted) {
    if (!w || (!data && len)) return -1;
    return vrt_load_vox_memory(data, len, w->root, ox, oy, oz, inserted) ? 1 : 0;
}

int vrth_world_insert(vrth_world *w, int x, int y, int z, uint32_t rgba, float refraction, float illumination, float k) {
    if (!w) return -1;
    Voxel m = {refraction, illumination, k};
    octree_insert(w->root, VoxelObjCreate(m, rgba, iv3(x, y, z)));
    return 0;
}

int vrth_world_insert_many(vrth_world *w, const int32_t *xyz, const uint32_t *rgba, size_t n, float refraction,
                           float illumination, float k) {
    if (!w || !xyz || !rgba) return -1;
    Voxel m = {refraction, illumination, k};
    for (size_t i = 0; i < n; ++i) octree_insert(w->root, VoxelObjCreate(m, rgba[i], iv3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])));
    return 0;
}

int vrth_world_remove(vrth_world *w, int x, int y, int z) {
    if (!w) return -1;
    octree_remove(w->root, iv3(x, y, z));
    return 0;
}

int vrth_world_find(vrth_world *w, int x, int y, int z, uint32_t out7[7]) {
    if (!w || !out7) return -1;
    const Voxel_Object v = octree_find(w->root, iv3(x, y, z));
    out7[0] = (uint32_t)v.coord.x; out7[1] = (uint32_t)v.coord.y; out7[2] = (uint32_t)v.coord.z;
    out7[3] = v.color;
    out7[4] = fbits(v.voxel.refraction); out7[5] = fbits(v.voxel.illumination); out7[6] = fbits(v.voxel.k);
    return 0;
}

int vrth_world_ray_cast(vrth_world *w, const float origin[3], const float dir[3], int32_t hit_coord[3], int *has_voxel) {
    if (!w || !origin || !dir) return -1;
    Ray r;
    r.origin.x = origin[0]; r.origin.y = origin[1]; r.origin.z = origin[2];
    r.direction.x = dir[0]; r.direction.y = dir[1]; r.direction.z = dir[2];
    Vector3 lo, hi;
    lo.x = lo.y = lo.z = 0.0f;
    hi.x = hi.y = hi.z = 1024.0f;
    Octree *n = octree_ray_cast(w->root, r, lo, hi);
    if (!n) return 0;
    if (hit_coord) { hit_coord[0] = n->voxel.coord.x; hit_coord[1] = n->voxel.coord.y; hit_coord[2] = n->voxel.coord.z; }
    if (has_voxel) *has_voxel = n->has_voxel ? 1 : 0;
    return 1;
}

long vrth_world_ray_cast_many(vrth_world *w, const float origin[3], const float *dirs, size_t n, uint8_t *hit, int32_t *hit_coords) {
    if (!w || !origin || !dirs) return -1;
    long hits = 0;
    for (size_t i = 0; i < n; ++i) {
        int32_t c[3] = {0, 0, 0};
        int has = 0;
        const int r = vrth_world_ray_cast(w, origin, dirs + 3 * i, c, &has);
        if (hit) hit[i] = (uint8_t)(r == 1 ? (has ? 1 : 2) : 0);
        if (hit_coords) { hit_coords[3 * i] = c[0]; hit_coords[3 * i + 1] = c[1]; hit_coords[3 * i + 2] = c[2]; }
        hits += r == 1;
    }
    return hits;
}

size_t vrth_world_texel_count(vrth_world *w) { return w ? _octree_texel_size(w->root) : 0; }

int vrth_world_flatten(vrth_world *w, uint8_t **texels, size_t *bytes, uint32_t *tex_dim) {
    if (!w || !texels || !bytes || !tex_dim) return -1;
    const size_t n = _octree_texel_size(w->root);
    size_t d = (size_t)ceil(cbrt((double)n));  // src/main.cpp:265-268
    if (d == 0) d = 1;
    *tex_dim = (uint32_t)d;
    size_t used = 0;
    *texels = octree_texture(w->root, &used, d);
    *bytes = *texels ? used : 0;
    return 0;
}

// The device record array (level order, 2 words per record; layout in voxel-raytracer_amd/csrc/vrt_layout.h)
// emitted straight from the pointer octree: what vrt_upload_octree() would derive from octree_texture()'s
// stream, without writing or re-parsing that stream. Same presence rule (a child counts when it holds a
// voxel or has children), same leaf bytes ((u8)(refraction*85), (u8)(illumination*255), (u8)(k*255)), same
// 16-level cut-off. Returns -2 for a tree whose root is itself a leaf (the stream form of that tree is
// read by the shader as a header; use the texel path for it).
int vrth_world_records(vrth_world *w, uint32_t **records, size_t *n_records, uint32_t *tex_dim) {
    if (!w) return -1;
    return vrth_octree_records(w->root, records, n_records, tex_dim);
}

namespace {
// a child the flattening counts (src/octree.cpp:524-545): it holds a voxel or has a children array
inline bool counted(const Octree *c) { return c && (c->has_voxel || c->children); }

// Records of the sub-tree under `top` in level order, top = record 0 (child indices local to the sub-tree);
// `top_depth` = depth of `top` below the root, for the 16-iteration truncation of the shader's descent.
// box != NULL ({lo x, y, z, hi x, y, z}, inclusive voxel coordinates: the region an edit touched): only children whose cube
// meets that box are expanded; every other internal child becomes a "keep" record {0xffffffff, 0xffffffff} (the child and
// everything below it is as it was: vrt_patch_apply shares it).
void emit_records(const Octree *top, uint32_t top_depth, std::vector<uint32_t> &out, const int *box = nullptr) {
    struct Item { const Octree *n; uint32_t rec, depth; };
    out.assign(2, 0u);
    std::vector<Item> queue;
    queue.push_back(Item{top, 0u, top_depth});
    for (size_t head = 0; head < queue.size(); ++head) {
        const Item it = queue[head];
        uint32_t mask = 0, leaf_mask = 0;
        if (it.n->children && it.depth < 15)
            for (int i = 0; i < 8; ++i)
                if (counted(it.n->children[i])) mask |= 1u << i;
        const uint32_t first = (uint32_t)(out.size() / 2);
        for (int i = 0; i < 8; ++i) {
            if (!(mask & (1u << i))) continue;
            const Octree *c = it.n->children[i];
            const uint32_t idx = (uint32_t)(out.size() / 2);
            if (!c->children && c->has_voxel) {
                const ColorRGBA col = c->voxel.color;
                const uint32_t w0 = (uint32_t)get_red_rgba(col) | ((uint32_t)get_green_rgba(col) << 8) |
                                    ((uint32_t)get_blue_rgba(col) << 16) | ((uint32_t)get_alpha_rgba(col) << 24);
                const uint32_t w1 = (uint32_t)(uint8_t)(c->voxel.voxel.refraction * 85.0f) |
                                    ((uint32_t)(uint8_t)(c->voxel.voxel.illumination * 255.0f) << 8) |
                                    ((uint32_t)(uint8_t)(c->voxel.voxel.k * 255.0f) << 16);
                out.push_back(w0);
                out.push_back(w1);
                leaf_mask |= 1u << i;
            } else if (box && !(box[3] >= c->left_bot_back.x && box[0] < c->right_top_front.x &&
                                box[4] >= c->left_bot_back.y && box[1] < c->right_top_front.y &&
                                box[5] >= c->left_bot_back.z && box[2] < c->right_top_front.z)) {
                out.push_back(0xffffffffu);
                out.push_back(0xffffffffu);
            } else {
                out.push_back(0u);
                out.push_back(0u);
                queue.push_back(Item{c, idx, it.depth + 1});
            }
        }
        out[2 * (size_t)it.rec] = mask | (leaf_mask << 8);
        out[2 * (size_t)it.rec + 1] = first;
    }
}

int hand_over(const std::vector<uint32_t> &out, uint32_t **records, size_t *n_records) {
    *n_records = out.size() / 2;
    *records = (uint32_t *)malloc(out.size() * sizeof(uint32_t));
    if (!*records) return -1;
    memcpy(*records, out.data(), out.size() * sizeof(uint32_t));
    return 0;
}

// the node reached from the root by `depth` child indices, or NULL when the flattening would not reach it
const Octree *walk(const Octree *root, const uint8_t *path, int depth) {
    const Octree *n = root;
    for (int d = 0; d < depth; ++d) {
        if (!n->children || path[d] > 7 || d >= 15) return nullptr;
        const Octree *c = n->children[path[d]];
        if (!counted(c)) return nullptr;
        n = c;
    }
    return n;
}
}  // namespace

// the same for a C++ caller that holds the Octree* itself (the reference's chunk0)
int vrth_octree_records(void *octree_root, uint32_t **records, size_t *n_records, uint32_t *tex_dim) {
    if (!octree_root || !records || !n_records || !tex_dim) return -1;
    Octree *root = static_cast<Octree *>(octree_root);
    const size_t texels = _octree_texel_size(root);
    size_t d = (size_t)ceil(cbrt((double)texels));
    *tex_dim = (uint32_t)(d == 0 ? 1 : d);
    if (!root->children && root->has_voxel) return -2;
    std::vector<uint32_t> out;
    emit_records(root, 0u, out);
    return hand_over(out, records, n_records);
}

// 0: the flattened tree has no node there, 1: a leaf, 2: an internal node
int vrth_octree_node_state(void *octree_root, const uint8_t *path, int depth) {
    if (!octree_root || (!path && depth > 0) || depth < 0 || depth > 15) return -1;
    const Octree *n = walk(static_cast<const Octree *>(octree_root), path, depth);
    if (!n) return 0;
    if (!n->children) return n->has_voxel ? 1 : 0;
    return 2;
}

int vrth_octree_subtree_records(void *octree_root, const uint8_t *path, int depth, uint32_t **records, size_t *n_records) {
    if (!octree_root || (!path && depth > 0) || depth < 0 || depth > 15 || !records || !n_records) return -1;
    const Octree *n = walk(static_cast<const Octree *>(octree_root), path, depth);
    if (!n || !n->children) return -2;
    std::vector<uint32_t> out;
    emit_records(n, (uint32_t)depth, out);
    return hand_over(out, records, n_records);
}

// the same sub-tree for an edit of voxel (x, y, z): only the nodes that contain the voxel are walked
int vrth_octree_path_records(void *octree_root, const uint8_t *path, int depth, int x, int y, int z, uint32_t **records,
                             size_t *n_records) {
    if (!octree_root || (!path && depth > 0) || depth < 0 || depth > 15 || !records || !n_records) return -1;
    const Octree *n = walk(static_cast<const Octree *>(octree_root), path, depth);
    if (!n || !n->children) return -2;
    const int box[6] = {x, y, z, x, y, z};
    std::vector<uint32_t> out;
    emit_records(n, (uint32_t)depth, out, box);
    return hand_over(out, records, n_records);
}

// ... and for an edit of a whole box of voxels [lo, hi] (inclusive): one sub-tree, the nodes that meet the box walked and emitted
int vrth_octree_box_records(void *octree_root, const uint8_t *path, int depth, const int32_t lo[3], const int32_t hi[3], uint32_t **records,
                            size_t *n_records) {
    if (!octree_root || (!path && depth > 0) || depth < 0 || depth > 15 || !records || !n_records || !lo || !hi) return -1;
    const Octree *n = walk(static_cast<const Octree *>(octree_root), path, depth);
    if (!n || !n->children) return -2;
    const int box[6] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    if (box[0] > box[3] || box[1] > box[4] || box[2] > box[5]) return -1;
    std::vector<uint32_t> out;
    emit_records(n, (uint32_t)depth, out, box);
    return hand_over(out, records, n_records);
}

int vrth_world_box_records(vrth_world *w, const uint8_t *path, int depth, const int32_t lo[3], const int32_t hi[3], uint32_t **records,
                           size_t *n_records) {
    return w ? vrth_octree_box_records(w->root, path, depth, lo, hi, records, n_records) : -1;
}

int vrth_world_path_records(vrth_world *w, const uint8_t *path, int depth, int x, int y, int z, uint32_t **records,
                            size_t *n_records) {
    return w ? vrth_octree_path_records(w->root, path, depth, x, y, z, records, n_records) : -1;
}

int vrth_world_node_state(vrth_world *w, const uint8_t *path, int depth) {
    return w ? vrth_octree_node_state(w->root, path, depth) : -1;
}

int vrth_world_subtree_records(vrth_world *w, const uint8_t *path, int depth, uint32_t **records, size_t *n_records) {
    return w ? vrth_octree_subtree_records(w->root, path, depth, records, n_records) : -1;
}

void vrth_free(void *p) { free(p); }

int vrth_camera_block(const float pos[3], float yaw, float pitch, int width, int height, float inv_projection[16],
                      float inv_view[16], float camera_pos[4], float *front3) {
    if (!pos || !inv_projection || !inv_view || !camera_pos || width < 1 || height < 1) return -1;
    Camera cam(vrtm::vec3(pos[0], pos[1], pos[2]), vrtm::vec3(0.0f, 1.0f, 0.0f), yaw, pitch);
    cam.FillDispatchBlock(width, height, inv_projection, inv_view, camera_pos);
    if (front3) { front3[0] = cam.Front.x; front3[1] = cam.Front.y; front3[2] = cam.Front.z; }
    return 0;
}

int vrth_encode_vox(int sx, int sy, int sz, const uint8_t *xyzi, size_t n, const uint8_t *palette, uint8_t **out, size_t *out_len) {
    if (!out || !out_len || (!xyzi && n) || sx < 1 || sy < 1 || sz < 1) return -1;
    std::vector<uint8_t> body;
    put_tag(body, "SIZE"); put_u32(body, 12); put_u32(body, 0);
    put_u32(body, (uint32_t)sx); put_u32(body, (uint32_t)sy); put_u32(body, (uint32_t)sz);
    put_tag(body, "XYZI"); put_u32(body, (uint32_t)(4 + 4 * n)); put_u32(body, 0);
    put_u32(body, (uint32_t)n);
    body.insert(body.end(), xyzi, xyzi + 4 * n);
    if (palette) {
        put_tag(body, "RGBA"); put_u32(body, 1024); put_u32(body, 0);
        body.insert(body.end(), palette, palette + 1024);
    }
    std::vector<uint8_t> file;
    put_tag(file, "VOX "); put_u32(file, 150);
    put_tag(file, "MAIN"); put_u32(file, 0); put_u32(file, (uint32_t)body.size());
    file.insert(file.end(), body.begin(), body.end());
    // the reference's chunk loop stops 12 bytes before the end of the file (src/voxReader.cpp:256),
    // so a chunk header must never be the last thing it needs; pad like exporters' trailing chunks do
    *out = (uint8_t *)malloc(file.size());
    if (!*out) return -1;
    memcpy(*out, file.data(), file.size());
    *out_len = file.size();
    return 0;
}

int vrth_write_vox(const char *path, int sx, int sy, int sz, const uint8_t *xyzi, size_t n, const uint8_t *palette) {
    uint8_t *buf = NULL;
    size_t len = 0;
    if (!path || vrth_encode_vox(sx, sy, sz, xyzi, n, palette, &buf, &len) != 0) return -1;
    FILE *fp = fopen(path, "wb");
    if (!fp) { free(buf); return -1; }
    const size_t w = fwrite(buf, 1, len, fp);
    fclose(fp);
    free(buf);
    return w == len ? 0 : -1;
}

// 64^3 model, floor slab z<4, solid sphere r=24 at (32,32,36);
// colorIndex = 1 + ((x*7 + y*13 + z*29) % 255); palette[i] = (i, 255-i, (i*37)&255, 255)
int vrth_make_custom_vox(uint8_t **out, size_t *out_len) {
    std::vector<uint8_t> xyzi;
    for (int z = 0; z < 64; ++z)
        for (int y = 0; y < 64; ++y)
            for (int x = 0; x < 64; ++x) {
                const int dx = x - 32, dy = y - 32, dz = z - 36;
                if (z < 4 || dx * dx + dy * dy + dz * dz <= 24 * 24) {
                    xyzi.push_back((uint8_t)x); xyzi.push_back((uint8_t)y); xyzi.push_back((uint8_t)z);
                    xyzi.push_back((uint8_t)(1 + ((x * 7 + y * 13 + z * 29) % 255)));
                }
            }
    uint8_t pal[1024];
    for (int i = 0; i < 256; ++i) { pal[4 * i] = (uint8_t)i; pal[4 * i + 1] = (uint8_t)(255 - i); pal[4 * i + 2] = (uint8_t)((i * 37) & 255); pal[4 * i + 3] = 255; }
    return vrth_encode_vox(64, 64, 64, xyzi.data(), xyzi.size() / 4, pal, out, out_len);
}

// BASELINE config 4: the reference's commented-out terrain generator (src/main.cpp:487-503) over a caller-supplied
// height field -- the loop order (z outer, x inner, y ascending), the material tests in the reference's order (the
// two lowest voxels of a column STONE, the top one DIRT, the rest GRASS; colours/materials of src/main.cpp:220-259),
// with the column's lower end at max(floor_y, h - band) as SURVEY.md 8(d) scales it (the reference fills from 20 up).
// heights: size_x * size_z uint16, row z, column x; columns x in [x0, x0 + nx), z in [z0, z0 + nz) are inserted at
// world (x, y, z). The noise itself is data: tests/golden/terrain_heights.npz holds the field the reference's own
// FastNoiseLite.h produces (generated by tests/golden/make_terrain.py).
int vrth_world_fill_heights(vrth_world *w, const uint16_t *heights, int size_x, int size_z, int x0, int z0, int nx, int nz,
                            int band, int floor_y) {
    if (!w || !heights || size_x < 1 || size_z < 1 || band < 1 || x0 < 0 || z0 < 0 || nx < 0 || nz < 0 ||
        x0 + nx > size_x || z0 + nz > size_z)
        return -1;
    for (int z = z0; z < z0 + nz; ++z)
        for (int x = x0; x < x0 + nx; ++x) {
            const int height = heights[(size_t)z * (size_t)size_x + (size_t)x];
            const int lo = height - band > floor_y ? height - band : floor_y;
            for (int h = lo; h < height; ++h) {
                const int kind = (h == lo || h == lo + 1) ? 5 : (h == height - 1 ? 1 : 0);  // VOX_STONE, VOX_DIRT, VOX_GRASS
                octree_insert(w->root, VoxelObjCreate(voxels[kind], voxelColors[kind], iv3(x, h, z)));
            }
        }
    return 0;
}

uint64_t vrth_fnv1a64(const uint8_t *p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

}  // extern "C"
