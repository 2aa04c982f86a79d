// voxReader.cpp -- MagicaVoxel .vox loader of the host API.
//
// Behaviour follows the reference's src/voxReader.cpp (cited inline): same chunk
// walk, same safety limits, same RAW fallback (files without a scene graph:
// every model's voxels inserted with the Y/Z swap) and the same scene-graph
// evaluation (nTRN translation * rotation, nGRP fan-out, nSHP centring by
// size/2 and half-away-from-zero rounding). The implementation parses the whole
// file from one in-memory buffer with explicit bounds handling.
#include <voxReader.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

// Material / colour tables of the reference application (src/main.cpp:220-259).
// Weak so that an application carrying its own definitions (as the reference's
// main.cpp does) overrides them.
__attribute__((weak)) Voxel voxels[] = {
    {3.0f, 0.0f, 0.0f},   // grass
    {3.0f, 0.0f, 0.0f},   // dirt
    {3.0f, 0.0f, 0.0f},   // wood
    {3.0f, 0.0f, 0.0f},   // leaves
    {1.33f, 0.0f, 0.0f},  // water
    {3.0f, 0.0f, 0.0f},   // stone
    {1.5f, 0.0f, 0.0f},   // glass
    {2.42f, 0.0f, 0.0f},  // diamond
    {1.38f, 0.0f, 0.0f},  // jelly
    {3.0f, 0.0f, 1.0f},   // mirror
    {3.0f, 1.0f, 0.0f},   // light
};
__attribute__((weak)) ColorRGBA voxelColors[] = {
    0x50b43cffu, 0x644628ffu, 0x78461effu, 0x1ea01effu, 0x3c64dc96u, 0xa0a0a0ffu,
    0xc8dcff50u, 0x00ffffffu, 0xff6464b4u, 0xffffffffu, 0xffd2d2ffu,
};

namespace {

constexpr int kSafeMin = -2048, kSafeMax = 2048;  // ref :18-19

bool verbose() {
    static const bool v = [] { const char *e = getenv("VRT_VERBOSE"); return e && *e && *e != '0'; }();
    return v;
}

// Sequential reader with stdio's short-read behaviour: a read past the end
// delivers the whole items that are available and leaves the rest untouched.
class Cursor {
public:
    Cursor(const uint8_t *p, size_t n) : p_(p), n_((long)n), pos_(0) {}
    long tell() const { return pos_; }
    void seek(long to) { pos_ = to; }
    void skip(long by) { pos_ += by; }
    long size() const { return n_; }
    size_t read(void *dst, size_t item, size_t count) {
        long avail = n_ - pos_;
        if (avail < 0) avail = 0;
        size_t items = item ? (size_t)avail / item : 0;
        if (items > count) items = count;
        memcpy(dst, p_ + pos_, items * item);
        pos_ += (long)(items * item);
        return items;
    }
    bool i32(int32_t &v) { return read(&v, 4, 1) == 1; }
    bool u8(uint8_t &v) { return read(&v, 1, 1) == 1; }

private:
    const uint8_t *p_;
    long n_, pos_;
};

typedef std::map<std::string, std::string> Dict;

std::string vox_string(Cursor &c) {  // ref :51-59
    int32_t len;
    if (!c.i32(len)) return "";
    if (len <= 0 || len > 1024 * 1024) return "";
    std::vector<char> buf((size_t)len + 1, '\0');
    if (c.read(buf.data(), 1, (size_t)len) != (size_t)len) return "";
    return std::string(buf.data());
}

Dict vox_dict(Cursor &c) {  // ref :61-72
    Dict d;
    int32_t pairs;
    if (!c.i32(pairs)) return d;
    if (pairs < 0 || pairs > 1000) return d;
    for (int i = 0; i < pairs; ++i) {
        std::string k = vox_string(c);
        std::string v = vox_string(c);
        d[k] = v;
    }
    return d;
}

struct Model {
    int32_t sx = 0, sy = 0, sz = 0;
    std::vector<uint8_t> xyzi;  // 4 bytes per voxel
    size_t n_voxels = 0;            // voxels the XYZI chunk claims; those past xyzi.size() / 4 read as zero bytes
    uint8_t byte(size_t i) const { return i < xyzi.size() ? xyzi[i] : (uint8_t)0; }
};

struct Node {
    enum Kind { Transform, Group, Shape } kind = Transform;
    int child = -1;
    float t[3] = {0.0f, 0.0f, 0.0f};
    uint8_t rot = 4;  // identity in the .vox encoding
    std::vector<int> kids;
    int model = -1;
};

struct Scene {
    std::vector<Model> models;
    std::map<int, Node> nodes;
    ColorRGBA palette[256];
};

// 4x4 column-major, products in glm's evaluation order
struct M4 {
    float m[16];
    static M4 identity() {
        M4 r;
        memset(r.m, 0, sizeof r.m);
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    float at(int c, int r) const { return m[c * 4 + r]; }
    float &at(int c, int r) { return m[c * 4 + r]; }
};

M4 mul(const M4 &a, const M4 &b) {
    M4 o;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            float s = a.at(0, r) * b.at(c, 0);
            s = s + a.at(1, r) * b.at(c, 1);
            s = s + a.at(2, r) * b.at(c, 2);
            s = s + a.at(3, r) * b.at(c, 3);
            o.at(c, r) = s;
        }
    return o;
}

M4 translation(const float t[3]) {  // glm::translate(mat4(1), t)
    M4 r = M4::identity();
    for (int k = 0; k < 3; ++k)
        r.at(3, k) = ((k == 0 ? 1.0f : 0.0f) * t[0] + (k == 1 ? 1.0f : 0.0f) * t[1]) + (k == 2 ? 1.0f : 0.0f) * t[2] + 0.0f;
    return r;
}

// .vox ROT byte: bits 0-1 / 2-3 = column of the non-zero entry in rows 0 / 1, bits 4-6 = row signs (ref :84-117)
M4 rotation(uint8_t rb) {
    const int r0 = rb & 3, r1 = (rb >> 2) & 3;
    float row0[3] = {0, 0, 0}, row1[3] = {0, 0, 0}, row2[3];
    if (r0 < 3) row0[r0] = (rb & 16) ? -1.0f : 1.0f;
    if (r1 < 3) row1[r1] = (rb & 32) ? -1.0f : 1.0f;
    row2[0] = row0[1] * row1[2] - row1[1] * row0[2];
    row2[1] = row0[2] * row1[0] - row1[2] * row0[0];
    row2[2] = row0[0] * row1[1] - row1[0] * row0[1];
    if (rb & 64) { row2[0] = -row2[0]; row2[1] = -row2[1]; row2[2] = -row2[2]; }
    M4 r = M4::identity();
    for (int c = 0; c < 3; ++c) { r.at(c, 0) = row0[c]; r.at(c, 1) = row1[c]; r.at(c, 2) = row2[c]; }
    return r;
}

inline int round_half_away(float v) { return v >= 0.0f ? (int)(v + 0.5f) : (int)(v - 0.5f); }  // ref :75-81

inline bool in_safe_box(int x, int y, int z) {
    return x >= kSafeMin && x <= kSafeMax && y >= kSafeMin && y <= kSafeMax && z >= kSafeMin && z <= kSafeMax;
}

inline ColorRGBA palette_entry(const Scene &s, uint8_t color_index) {
    int ci = (int)color_index - 1;
    if (ci < 0 || ci >= 256) ci = 0;
    return s.palette[ci];
}

// ref :121-211
// depth: a scene graph whose nTRN child or nGRP kid refers back to an ancestor would recurse until the host stack
// overflows (the reference does); no real file nests deeper than a handful of levels
constexpr int kMaxGraphDepth = 64;
void walk(const Scene &s, int id, const M4 &xf, Octree *tree, const int origin[3], const Voxel &material, int depth = 0) {
    if (depth > kMaxGraphDepth) {
        std::cerr << "vox: scene graph deeper than " << kMaxGraphDepth << " levels (a cycle?): branch dropped" << std::endl;
        return;
    }
    std::map<int, Node>::const_iterator it = s.nodes.find(id);
    if (it == s.nodes.end()) return;
    const Node &n = it->second;
    switch (n.kind) {
        case Node::Transform:
            walk(s, n.child, mul(mul(xf, translation(n.t)), rotation(n.rot)), tree, origin, material, depth + 1);
            break;
        case Node::Group:
            for (size_t i = 0; i < n.kids.size(); ++i) walk(s, n.kids[i], xf, tree, origin, material, depth + 1);
            break;
        case Node::Shape: {
            if (n.model < 0 || n.model >= (int)s.models.size()) return;
            const Model &m = s.models[(size_t)n.model];
            const float cx = (float)m.sx / 2.0f, cy = (float)m.sy / 2.0f, cz = (float)m.sz / 2.0f;
            for (size_t v = 0; v < m.n_voxels; ++v) {
                const size_t i = v * 4;
                const float lx = (float)m.byte(i) - cx, ly = (float)m.byte(i + 1) - cy, lz = (float)m.byte(i + 2) - cz;
                float w[3];
                for (int r = 0; r < 3; ++r)
                    w[r] = (xf.at(0, r) * lx + xf.at(1, r) * ly) + (xf.at(2, r) * lz + xf.at(3, r) * 1.0f);
                // MagicaVoxel is Z-up, the engine Y-up
                const int fx = origin[0] + round_half_away(w[0]);
                const int fy = origin[1] + round_half_away(w[2]);
                const int fz = origin[2] + round_half_away(w[1]);
                if (!in_safe_box(fx, fy, fz)) continue;
                IVector3 c;
                c.x = fx; c.y = fy; c.z = fz;
                octree_insert(tree, VoxelObjCreate(material, palette_entry(s, m.byte(i + 3)), c));
            }
            break;
        }
    }
}

// ref :255-377
void parse_chunks(Cursor &c, Scene &s) {
    int32_t last_size[3] = {0, 0, 0};
    const long file_size = c.size();
    c.seek(8);
    while (c.tell() < file_size - 12) {
        char id[4];
        int32_t content, children;
        if (c.read(id, 1, 4) < 4 || !c.i32(content) || !c.i32(children)) break;
        if (content < 0 || children < 0) {
            std::cerr << "vox: negative chunk size" << std::endl;
            break;
        }
        const long end = c.tell() + content + children;
        if (end > file_size) {
            std::cerr << "vox: chunk runs past the end of the file" << std::endl;
            break;
        }
        if (!memcmp(id, "MAIN", 4)) continue;  // its children follow inline
        if (!memcmp(id, "PACK", 4)) {
            c.skip(content);
        } else if (!memcmp(id, "SIZE", 4)) {
            c.i32(last_size[0]); c.i32(last_size[1]); c.i32(last_size[2]);
        } else if (!memcmp(id, "XYZI", 4)) {
            int32_t count = 0;
            c.i32(count);
            if (count < 0 || count > 10000000) {
                std::cerr << "vox: implausible voxel count " << count << std::endl;
                c.seek(end);
                continue;
            }
            Model m;
            m.sx = last_size[0]; m.sy = last_size[1]; m.sz = last_size[2];
            // The reference callocs count * 4 bytes and freads what the file holds (src/voxReader.cpp:284-297): a chunk that
            // claims more voxels than it carries leaves the rest zero. Only the bytes that exist are stored here -- the
            // zeros are implied by n_voxels -- so a small file with many inflated counts cannot exhaust memory.
            m.n_voxels = (size_t)count;
            const long have = file_size - c.tell();
            const size_t stored = have <= 0 ? 0 : ((size_t)count * 4 < (size_t)have ? (size_t)count * 4 : (size_t)have);
            m.xyzi.assign(stored, 0);
            for (size_t i = 0; i < m.xyzi.size(); ++i) c.u8(m.xyzi[i]);
            s.models.push_back(m);
        } else if (!memcmp(id, "RGBA", 4)) {
            for (int i = 0; i < 256; ++i) {
                uint8_t q[4] = {0, 0, 0, 0};
                c.u8(q[0]); c.u8(q[1]); c.u8(q[2]); c.u8(q[3]);
                s.palette[i] = make_color_rgba(q[0], q[1], q[2], q[3]);
            }
        } else if (!memcmp(id, "nTRN", 4)) {
            Node n;
            n.kind = Node::Transform;
            int32_t nid = 0, reserved, layer, frames = 0;
            c.i32(nid);
            vox_dict(c);
            int32_t child = -1;
            c.i32(child);
            n.child = child;
            c.i32(reserved); c.i32(layer); c.i32(frames);
            for (int f = 0; f < frames; ++f) {
                Dict d = vox_dict(c);
                if (f != 0) continue;
                Dict::iterator t = d.find("_t"), r = d.find("_r");
                if (t != d.end()) {  // three whitespace-separated numbers; stops at the first that fails
                    const char *p = t->second.c_str();
                    for (int k = 0; k < 3; ++k) {
                        char *e;
                        const float v = strtof(p, &e);
                        if (e == p) break;
                        n.t[k] = v;
                        p = e;
                    }
                }
                if (r != d.end()) n.rot = (uint8_t)atoi(r->second.c_str());
            }
            s.nodes[nid] = n;
        } else if (!memcmp(id, "nGRP", 4)) {
            Node n;
            n.kind = Node::Group;
            int32_t nid = 0, kids = 0;
            c.i32(nid);
            vox_dict(c);
            c.i32(kids);
            for (int i = 0; i < kids; ++i) {
                int32_t k = 0;
                if (!c.i32(k)) break;
                n.kids.push_back(k);
            }
            s.nodes[nid] = n;
        } else if (!memcmp(id, "nSHP", 4)) {
            Node n;
            n.kind = Node::Shape;
            int32_t nid = 0, count = 0;
            c.i32(nid);
            vox_dict(c);
            c.i32(count);
            for (int i = 0; i < count && c.tell() < file_size; ++i) {
                int32_t mid = 0;
                c.i32(mid);
                vox_dict(c);
                if (i == 0) n.model = mid;
            }
            s.nodes[nid] = n;
        }
        c.seek(end);
    }
}

}  // namespace

// In-memory entry point (also the body of load_vox_file). *inserted (optional)
// receives the number of octree_insert calls made.
bool vrt_load_vox_memory(const uint8_t *data, size_t len, Octree *tree, int offsetX, int offsetY, int offsetZ,
                         long *inserted) {
    if (inserted) *inserted = 0;
    if (!tree) {
        std::cerr << "vox: octree is NULL" << std::endl;
        return false;
    }
    Cursor c(data, len);
    char magic[4];
    int32_t version;
    if (c.read(magic, 1, 4) != 4 || !c.i32(version)) return false;
    if (memcmp(magic, "VOX ", 4) != 0) {
        std::cerr << "vox: bad magic (not a .vox file)" << std::endl;
        return false;
    }
    Scene s;
    for (int i = 0; i < 256; ++i) s.palette[i] = make_color_rgba((uint8_t)i, (uint8_t)i, (uint8_t)i, 255);  // ref :244-246
    parse_chunks(c, s);

    const Voxel material = voxels[0];  // ref :21
    const int origin[3] = {offsetX, offsetY, offsetZ};
    if (s.nodes.empty()) {  // RAW mode, ref :382-408
        long count = 0;
        for (size_t mi = 0; mi < s.models.size(); ++mi) {
            const Model &v = s.models[mi];
            for (size_t k = 0; k < v.n_voxels; ++k) {
                const size_t i = k * 4;
                const int fx = offsetX + v.byte(i), fy = offsetY + v.byte(i + 2), fz = offsetZ + v.byte(i + 1);
                if (!in_safe_box(fx, fy, fz)) continue;
                IVector3 p;
                p.x = fx; p.y = fy; p.z = fz;
                octree_insert(tree, VoxelObjCreate(material, palette_entry(s, v.byte(i + 3)), p));
                ++count;
            }
        }
        if (verbose()) std::cout << "vox: no scene graph, loaded " << count << " voxels (raw mode)" << std::endl;
        if (inserted) *inserted = count;
        return count > 0;
    }
    if (s.nodes.count(0)) {  // ref :411-415
        if (verbose()) std::cout << "vox: walking scene graph (" << s.nodes.size() << " nodes)" << std::endl;
        walk(s, 0, M4::identity(), tree, origin, material);
    }
    return true;
}

// ref :215-418
bool load_vox_file(const char *filename, Octree *tree, int offsetX, int offsetY, int offsetZ) {
    if (!tree) {
        std::cerr << "vox: octree is NULL" << std::endl;
        return false;
    }
    FILE *fp = filename ? fopen(filename, "rb") : NULL;
    if (!fp) {
        std::cerr << "vox: cannot open " << (filename ? filename : "(null)") << std::endl;
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(fp);
    return vrt_load_vox_memory(buf.data(), buf.size(), tree, offsetX, offsetY, offsetZ, NULL);
}
