// octree.cpp -- host-side sparse voxel octree: build, edit, CPU ray cast and
// flatten to the texel stream the ray-casting layer uploads.
//
// API and observable behaviour follow the reference's src/octree.cpp (cited per
// function); the implementation is this repository's: the eight children of a
// node live in ONE allocation (pointer table + node storage), insert/remove walk
// the tree iteratively with an explicit path, and the flatten is a sizing pass
// plus an emit pass over that structure. octree_texture() must stay
// byte-identical to the reference's, including two behaviours that look like
// bugs but shape the stream (tests/golden/flatten.json pins them):
//   * splitting marks one child as holding the parent's voxel even when the
//     parent held none ("phantom" alpha-0 leaves)            ref :174-179
//   * the merged-volume test uses vmm's ivec3_equal_vec, whose shipped binary
//     ignores y except for being non-zero                     ref :227
#include <octree.hpp>

#include <math.h>
#include <stdio.h>
#include <string.h>

namespace {

constexpr int kMinHeight = -1024;  // ref :12-14 (MIN_HEIGHT)
constexpr int kMaxPath = 72;

// One allocation per split: `slots` is what Octree::children points at.
struct ChildBlock {
    Octree *slots[8];
    Octree nodes[8];
};

inline Voxel_Object no_voxel() {  // ref :36-40
    Voxel_Object v;
    memset(&v, 0, sizeof v);
    v.coord.y = kMinHeight;
    return v;
}

inline IVector3 iv(int x, int y, int z) {
    IVector3 r;
    r.x = x; r.y = y; r.z = z;
    return r;
}

inline IVector3 split_point(const IVector3 &lo, const IVector3 &hi) {
    return iv(lo.x + (hi.x - lo.x) / 2, lo.y + (hi.y - lo.y) / 2, lo.z + (hi.z - lo.z) / 2);
}

// child index: x high -> 4, y high -> 2, z high -> 1 (ref :46-76)
inline int octant(const IVector3 &c, const IVector3 &mid) {
    return (c.x >= mid.x ? 4 : 0) | (c.y >= mid.y ? 2 : 0) | (c.z >= mid.z ? 1 : 0);
}

inline bool outside(const IVector3 &c, const IVector3 &lo, const IVector3 &hi) {  // ref :80-87
    return c.x < lo.x || c.x >= hi.x || c.y < lo.y || c.y >= hi.y || c.z < lo.z || c.z >= hi.z;
}

inline bool unit_cell(const Octree *n) {
    return n->right_top_front.x - n->left_bot_back.x <= 1 && n->right_top_front.y - n->left_bot_back.y <= 1 &&
           n->right_top_front.z - n->left_bot_back.z <= 1;
}

inline bool holds_leaf(const Octree *n) { return n && n->has_voxel && !n->children; }  // ref :185-187

inline bool occupied(const Octree *n) { return n && (n->has_voxel || n->children); }

bool in_parent_block(const Octree *n) {
    if (!n->parent || !n->parent->children) return false;
    const ChildBlock *b = reinterpret_cast<const ChildBlock *>(n->parent->children);
    return n >= &b->nodes[0] && n <= &b->nodes[7];
}

void release_children(Octree *n) {
    free(n->children);  // the ChildBlock
    n->children = NULL;
}

// ref :132-182 (_create_children)
bool make_children(Octree *n) {
    ChildBlock *b = static_cast<ChildBlock *>(calloc(1, sizeof(ChildBlock)));
    if (!b) return false;
    const IVector3 lo = n->left_bot_back, hi = n->right_top_front;
    const IVector3 mid = split_point(lo, hi);
    for (int i = 0; i < 8; ++i) {
        Octree *c = &b->nodes[i];
        c->parent = n;
        c->left_bot_back = iv((i & 4) ? mid.x : lo.x, (i & 2) ? mid.y : lo.y, (i & 1) ? mid.z : lo.z);
        c->right_top_front = iv((i & 4) ? hi.x : mid.x, (i & 2) ? hi.y : mid.y, (i & 1) ? hi.z : mid.z);
        c->voxel = no_voxel();
        b->slots[i] = c;
    }
    n->children = b->slots;
    // whatever the node's voxel slot held moves to the octant of its coord, and that child is
    // flagged occupied unconditionally
    Octree *heir = b->slots[octant(n->voxel.coord, mid)];
    heir->voxel = n->voxel;
    heir->has_voxel = true;
    n->voxel = no_voxel();
    n->has_voxel = false;
    return true;
}

// ref :203-255 (_split_node)
bool split(Octree *n) {
    if (n->children) return true;
    const Voxel_Object held = n->voxel;
    const bool was_solid = n->has_voxel;
    if (!make_children(n)) return false;
    if (was_solid) {
        if (ivec3_equal_vec(held.coord, n->left_bot_back)) {
            // a merged volume: every octant inherits the material at its own corner
            for (int i = 0; i < 8; ++i) {
                Octree *c = n->children[i];
                c->voxel = held;
                c->voxel.coord = c->left_bot_back;
                c->has_voxel = true;
            }
        } else {
            Octree *c = n->children[octant(held.coord, split_point(n->left_bot_back, n->right_top_front))];
            c->voxel = held;
            c->has_voxel = true;
        }
        n->has_voxel = false;
    }
    return true;
}

// ref :190-200 (_nodes_are_identical)
bool same_material(const Octree *a, const Octree *b) {
    if (!holds_leaf(a) || !holds_leaf(b)) return false;
    const bool same = a->voxel.color == b->voxel.color && a->voxel.voxel.refraction == b->voxel.voxel.refraction &&
                      a->voxel.voxel.illumination == b->voxel.voxel.illumination;
    return same || (a->voxel.coord.y <= kMinHeight && b->voxel.coord.y <= kMinHeight);
}

// ref :258-285 (_try_merge_children)
void merge_if_uniform(Octree *n) {
    if (!n->children) return;
    for (int i = 0; i < 8; ++i)
        if (!holds_leaf(n->children[i])) return;
    for (int i = 1; i < 8; ++i)
        if (!same_material(n->children[0], n->children[i])) return;
    n->voxel = n->children[0]->voxel;
    n->voxel.coord = n->left_bot_back;
    n->has_voxel = true;
    release_children(n);
}

// ref :488-498 (_get_child_mask)
inline unsigned presence_mask(const Octree *n) {
    unsigned m = 0;
    if (n->children)
        for (int i = 0; i < 8; ++i)
            if (occupied(n->children[i])) m |= 1u << i;
    return m;
}

// ref :524-552
size_t measure(const Octree *n) {
    if (!n) return 0;
    if (!n->children) return n->has_voxel ? 2 : 0;
    const unsigned m = presence_mask(n);
    if (!m) return 0;
    size_t total = 1 + (size_t)__builtin_popcount(m);
    for (int i = 0; i < 8; ++i)
        if (m & (1u << i)) total += measure(n->children[i]);
    return total;
}

inline void put24(uint8_t *texel, size_t value, bool leaf_flag) {  // ref :556-570
    const uint32_t v = (uint32_t)value | (leaf_flag ? 0x800000u : 0u);
    texel[0] = (uint8_t)v;
    texel[1] = (uint8_t)(v >> 8);
    texel[2] = (uint8_t)(v >> 16);
}

// ref :573-655 -- pre-order: header, pointer list, then each present child's subtree
void emit(const Octree *n, uint8_t *tex, size_t &cursor) {
    if (!n->children) {
        if (!n->has_voxel) return;
        uint8_t *t = tex + cursor * 4;
        t[0] = get_red_rgba(n->voxel.color);
        t[1] = get_green_rgba(n->voxel.color);
        t[2] = get_blue_rgba(n->voxel.color);
        t[3] = 255;
        t[4] = (uint8_t)(n->voxel.voxel.refraction * 85.0f);
        t[5] = (uint8_t)(n->voxel.voxel.illumination * 255.0f);
        t[6] = (uint8_t)(n->voxel.voxel.k * 255.0f);
        t[7] = get_alpha_rgba(n->voxel.color);
        cursor += 2;
        return;
    }
    const unsigned m = presence_mask(n);
    if (!m) return;
    const size_t header = cursor++;
    const size_t list = cursor;
    cursor += (size_t)__builtin_popcount(m);
    put24(tex + header * 4, list, false);
    tex[header * 4 + 3] = (uint8_t)m;
    size_t slot = list;
    for (int i = 0; i < 8; ++i) {
        if (!(m & (1u << i))) continue;
        const Octree *c = n->children[i];
        put24(tex + slot * 4, cursor, c->children == NULL && c->has_voxel);
        ++slot;
        emit(c, tex, cursor);
    }
}

// ref :364-403 (_octree_find_leaf)
Octree *locate(Octree *root, const IVector3 &p, IVector3 *nmin, IVector3 *nmax) {
    IVector3 lo = root->left_bot_back, hi = root->right_top_front;
    if (outside(p, lo, hi)) return NULL;
    Octree *cur = root;
    while (cur->children) {
        const IVector3 mid = split_point(lo, hi);
        const int ci = octant(p, mid);
        if (ci & 4) lo.x = mid.x; else hi.x = mid.x;
        if (ci & 2) lo.y = mid.y; else hi.y = mid.y;
        if (ci & 1) lo.z = mid.z; else hi.z = mid.z;
        cur = cur->children[ci];
        if (!cur) break;
    }
    *nmin = lo;
    *nmax = hi;
    return cur;
}

}  // namespace

Octree *octree_new(void) { return static_cast<Octree *>(calloc(1, sizeof(Octree))); }

// ref :93-100
Octree *octree_create(Octree *parent, IVector3 left_bot_back, IVector3 right_top_front) {
    Octree *n = octree_new();
    if (!n) return NULL;
    n->parent = parent;
    n->left_bot_back = left_bot_back;
    n->right_top_front = right_top_front;
    return n;
}

// ref :287-323
void octree_insert(Octree *tree, Voxel_Object voxel) {
    if (!tree || outside(voxel.coord, tree->left_bot_back, tree->right_top_front)) return;
    Octree *path[kMaxPath];
    int depth = 0;
    Octree *n = tree;
    while (n && depth < kMaxPath) {
        if (unit_cell(n)) {
            n->voxel = voxel;
            n->has_voxel = true;
            break;
        }
        if (!n->children && !split(n)) break;
        path[depth++] = n;
        n = n->children[octant(voxel.coord, split_point(n->left_bot_back, n->right_top_front))];
    }
    while (depth > 0) merge_if_uniform(path[--depth]);
}

// ref :684-740
void octree_remove(Octree *tree, IVector3 coord) {
    if (!tree || outside(coord, tree->left_bot_back, tree->right_top_front)) return;
    Octree *path[kMaxPath];
    int depth = 0;
    Octree *n = tree;
    while (n && depth < kMaxPath) {
        if (unit_cell(n)) {
            n->has_voxel = false;
            break;
        }
        if (!n->children && n->has_voxel && !split(n)) break;
        if (!n->children) break;  // air below here
        path[depth++] = n;
        n = n->children[octant(coord, split_point(n->left_bot_back, n->right_top_front))];
    }
    while (depth > 0) {
        Octree *p = path[--depth];
        bool any = false;
        for (int i = 0; i < 8 && !any; ++i) any = occupied(p->children[i]);
        if (!any) {
            release_children(p);
            p->has_voxel = false;
        }
    }
}

// ref :102-130 -- note the (lo+hi)/2 midpoint here, unlike insert/flatten
Voxel_Object octree_find(Octree *tree, IVector3 coord) {
    for (Octree *n = tree; n;) {
        if (outside(coord, n->left_bot_back, n->right_top_front)) break;
        if (n->has_voxel && ivec3_equal_vec(n->voxel.coord, coord)) return n->voxel;
        if (!n->children) break;
        const IVector3 mid = ivec3_scalar_div(ivec3_add(n->left_bot_back, n->right_top_front), 2);
        n = n->children[octant(coord, mid)];
    }
    return no_voxel();
}

// ref :405-485
Octree *octree_ray_cast(Octree *root, Ray ray, Vector3 box_min, Vector3 box_max) {
    if (!root) return NULL;
    float px = ray.origin.x, py = ray.origin.y, pz = ray.origin.z;
    const float dx = ray.direction.x, dy = ray.direction.y, dz = ray.direction.z;
    const float ix = fabsf(dx) < 1e-8f ? 1e20f : 1.0f / dx;
    const float iy = fabsf(dy) < 1e-8f ? 1e20f : 1.0f / dy;
    const float iz = fabsf(dz) < 1e-8f ? 1e20f : 1.0f / dz;
    IVector3 cell = iv((int)floorf(px), (int)floorf(py), (int)floorf(pz));
    IVector3 nmin = ivec3_vec3(box_min), nmax = ivec3_vec3(box_max);
    for (int step = 0; step < 512; ++step) {
        Octree *n = locate(root, cell, &nmin, &nmax);
        if (n && n->has_voxel && n->voxel.coord.y > kMinHeight) return n;
        const float tx = (dx > 0.0f ? (float)nmax.x - px : (float)nmin.x - px) * ix;
        const float ty = (dy > 0.0f ? (float)nmax.y - py : (float)nmin.y - py) * iy;
        const float tz = (dz > 0.0f ? (float)nmax.z - pz : (float)nmin.z - pz) * iz;
        const float tyz = ty < tz ? ty : tz;
        float t = tx < tyz ? tx : tyz;
        const int axis = (tx < ty) ? ((tx < tz) ? 0 : 2) : ((ty < tz) ? 1 : 2);
        if (t < 0.0001f) t = 0.0001f;
        px += dx * t; py += dy * t; pz += dz * t;
        float qx = px, qy = py, qz = pz;
        if (axis == 0) qx += dx * 0.001f;
        else if (axis == 1) qy += dy * 0.001f;
        else qz += dz * 0.001f;
        cell = iv((int)floorf(qx), (int)floorf(qy), (int)floorf(qz));
        if (outside(cell, root->left_bot_back, root->right_top_front)) return NULL;
    }
    return NULL;
}

size_t _octree_texel_size(Octree *tree) { return measure(tree); }

// ref :657-682
uint8_t *octree_texture(Octree *tree, size_t *arr_size, size_t tex_dim) {
    (void)tex_dim;  // the stream is linear; the dimension only matters to the 3-D packaging
    if (!tree || !arr_size) return NULL;
    const size_t texels = measure(tree);
    if (texels == 0) {
        *arr_size = 0;
        return NULL;
    }
    *arr_size = texels * 4;
    uint8_t *tex = static_cast<uint8_t *>(calloc(texels * 4, 1));
    if (!tex) return NULL;
    size_t cursor = 0;
    emit(tree, tex, cursor);
    if (cursor != texels) fprintf(stderr, "octree_texture: wrote %zu texels, sized %zu\n", cursor, texels);
    return tex;
}

// ref :743-754
void octree_delete(Octree *tree) {
    if (!tree) return;
    if (tree->children) {
        for (int i = 0; i < 8; ++i) {
            Octree *c = tree->children[i];
            if (c && c->children) octree_delete(c);
        }
        release_children(tree);
    }
    if (!in_parent_block(tree)) free(tree);
}

// ---- voxel value helpers (include/voxel.hpp) -------------------------------------------------------------------
Voxel_Object VoxelObjCreate(Voxel voxel, ColorRGBA color, IVector3 coord) { return Voxel_Object{coord, color, voxel}; }

bool voxel_compare(Voxel a, Voxel b) { return a.refraction == b.refraction && a.illumination == b.illumination; }  // k: not compared

bool voxel_obj_compare(Voxel_Object a, Voxel_Object b) { return ivec3_equal_vec(a.coord, b.coord) && voxel_compare(a.voxel, b.voxel); }
