"""The oracle against the golden vectors (CPU)."""
import os
import struct

import numpy as np

from conftest import MAPS


def _f(u):
    return struct.unpack("f", struct.pack("I", u))[0]


def test_flatten_matches_recorded_hashes(O, golden):
    for name, g in golden["flatten"]["maps"].items():
        tree, ok, n = O.load_vox(os.path.join(MAPS, name + ".vox"))
        assert ok and n == g["voxels_inserted"]
        tex, dim = O.flatten(tree)
        assert tex.size == g["bytes"] and dim == g["tex_dim"]
        assert "%016x" % O.fnv1a64(tex) == g["fnv1a64"]
        # root header: pointer list right behind it
        assert int(tex[0]) | int(tex[1]) << 8 | int(tex[2]) << 16 == 1
        O.lib().o_octree_delete(tree)


def test_camera_matches_reference_bits(O, golden):
    for c in golden["camera"]["cases"]:
        (ip, iv, cp), cam = O.camera_ubo([_f(x) for x in c["pos"]], _f(c["yaw"]), _f(c["pitch"]), c["width"], c["height"])
        assert list(ip.view(np.uint32)) == c["inv_proj"]
        assert list(iv.view(np.uint32)) == c["inv_view"]
        assert list(np.array(cam.front, np.float32).view(np.uint32)) == c["front"]
        assert list(np.array(cam.right, np.float32).view(np.uint32)) == c["right"]
        assert list(np.array(cam.up, np.float32).view(np.uint32)) == c["up"]
    s = O.Scene()
    O.lib().o_scene_defaults(s)
    assert list(np.array(s.light_dir, np.float32).view(np.uint32)) == golden["camera"]["light_dir"]


def test_small_frames_match_committed_hashes(O, golden):
    scenes = {}
    for key, g in golden["frames"]["frames"].items():
        if g["width"] * g["height"] > 64 * 1024:
            continue
        if g["map"] not in scenes:
            if g["map"] == "terrain_full":   # the whole field through the wide-pointer stream (extension, oracle.h); ~10 s
                t = golden["terrain"]
                tree = O.new_tree()
                O.fill_heights(tree, np.load(os.path.join(os.path.dirname(MAPS), "terrain_heights.npz"))["heights"],
                               0, 0, 1024, 1024, t["band"], t["floor"])
                scenes[g["map"]] = O.flatten(tree, wide=True)
                O.lib().o_octree_delete(tree)
            elif g["map"] == "room":
                from conftest import room_tree
                tree = room_tree(O)
            elif g["map"] == "terrain":   # BASELINE config 4: the height-field fixture (tests/golden/make_terrain.py)
                t = golden["terrain"]
                tree, wd = O.new_tree(), t["window"]
                O.fill_heights(tree, np.load(os.path.join(os.path.dirname(MAPS), "terrain_heights.npz"))["heights"],
                               wd["x0"], wd["z0"], wd["nx"], wd["nz"], t["band"], t["floor"])
            else:
                tree, ok, _ = O.load_vox(os.path.join(MAPS, g["map"] + ".vox"))
                assert ok, g["map"]
            if g["map"] not in scenes:
                scenes[g["map"]] = O.flatten(tree)
        tex, dim = scenes[g["map"]]
        p = g["pose"]
        (ip, iv, cp), _ = O.camera_ubo(p[:3], p[3], p[4], g["width"], g["height"])
        rgba, idd, _, st = O.render(O.make_scene(tex, dim, ip, iv, cp, wide=g["map"] == "terrain_full"), g["width"], g["height"], g["mode"])
        assert "%016x" % O.fnv1a64(rgba) == g["rgba_fnv1a64"], key
        assert "%016x" % O.fnv1a64(idd) == g["id_dist_fnv1a64"], key
        assert st["fetches"] == g["fetches"] and st["hits"] == g["hits"], key


def test_find_agrees_with_dense_grid(O):
    """octreeFind restatement vs a brute-force dense occupancy grid built from the voxel list."""
    rng = np.random.default_rng(7)
    L = O.lib()
    tree = O.new_tree()
    n = 3000
    xyz = rng.integers(0, 40, size=(n, 3))
    cols = rng.integers(1, 4, size=n)
    grid = {}
    for (x, y, z), c in zip(xyz, cols):
        color = (int(c) * 0x203040 << 8 | 0xff) & 0xffffffff
        L.o_octree_insert(tree, O.VoxelObj(O.IVec3(int(x), int(y), int(z)), color, O.Voxel(3.0, 0.0, 0.0)))
        grid[(int(x), int(y), int(z))] = color
    tex, dim = O.flatten(tree)
    s = O.make_scene(tex, dim, np.eye(4, dtype=np.float32).ravel(), np.eye(4, dtype=np.float32).ravel(), [0, 0, 0, 1])
    import ctypes as C
    leaf = (C.c_uint8 * 8)()
    mn, mx = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    for _ in range(4000):
        p = rng.integers(-3, 44, size=3)
        found = L.o_find_point(C.byref(s), (C.c_int32 * 3)(*[int(v) for v in p]), leaf, mn, mx)
        key = tuple(int(v) for v in p)
        assert all(mn[i] <= key[i] < mx[i] for i in range(3))
        if key in grid:
            assert found == 1 and leaf[7] == 255
            c = grid[key]
            assert (leaf[0], leaf[1], leaf[2]) == ((c >> 24) & 255, (c >> 16) & 255, (c >> 8) & 255)
        else:
            # empty space or a phantom alpha-0 leaf: never an opaque voxel
            assert found == 0 or leaf[7] == 0
        # the node found must not contain any OTHER material: every cell of an opaque leaf is that voxel
        if found == 1 and leaf[7] == 255:
            ext = [mx[i] - mn[i] for i in range(3)]
            if max(ext) <= 4:
                for dx in range(ext[0]):
                    for dy in range(ext[1]):
                        for dz in range(ext[2]):
                            assert (mn[0] + dx, mn[1] + dy, mn[2] + dz) in grid
    L.o_octree_delete(tree)


def test_leaf_texels_carry_the_reference_colour_getters(O, golden):
    """tests/golden/color.json is the output of the REFERENCE's src/color.c (oracle/ref_color_driver.c, built in place):
    the oracle's flattener must put get_red/green/blue_rgba into leaf texel 0 and get_alpha_rgba into texel 1
    (src/octree.cpp:587-596)."""
    L = O.lib()
    for c in golden["color"]["cases"]:
        tree = O.new_tree()
        L.o_octree_insert(tree, O.VoxelObj(O.IVec3(3, 5, 7), c["in"], O.Voxel(3.0, 0.0, 0.0)))
        tex, _ = O.flatten(tree)
        t = np.asarray(tex, np.uint8).reshape(-1, 4)
        leaves = [i for i in range(len(t) - 1) if t[i, 3] == 255 and t[i + 1, 0] == 255 and t[i + 1, 1] == 0 and t[i + 1, 2] == 0]
        assert leaves, c
        i = leaves[-1]          # the voxel's own leaf is the last one the depth-first writer emits on this path
        assert tuple(int(v) for v in t[i, :3]) == (c["get_red_rgba"], c["get_green_rgba"], c["get_blue_rgba"]), c
        assert int(t[i + 1, 3]) == c["get_alpha_rgba"], c
        L.o_octree_delete(tree)


def test_det_math_is_sane(O):
    L = O.lib()
    xs = np.linspace(-20, 5, 101)
    assert np.allclose([L.o_det_expf(float(x)) for x in xs], np.exp(xs), rtol=3e-7)
    ts = np.linspace(0, 6.3, 97)
    assert np.allclose([L.o_det_sinf(float(t)) for t in ts], np.sin(ts), atol=3e-7)
    assert np.allclose([L.o_det_cosf(float(t)) for t in ts], np.cos(ts), atol=3e-7)
    assert abs(L.o_det_powf(0.37, 5.0) - 0.37 ** 5) < 1e-8


def test_denoise_restatement_properties(O):
    """quad.frag restatement: sky passes through, a uniformly coloured voxel face keeps its colour, only same-id
    pixels mix, and the radius law clamp(int(200/sqrt(max(1,d))),1,20)."""
    H, W = 40, 60
    rgba = np.zeros((H, W, 4), np.uint8)
    rgba[..., 3] = 255
    idd = np.zeros((H, W, 2), np.int32)
    rgba[:, :30, :3] = (10, 200, 30)      # sky, id 0
    rgba[:, 30:, :3] = (90, 90, 90)
    idd[:, 30:, 0] = 7
    idd[:, 30:, 1] = 100                  # radius 20
    rgba[20, 45, :3] = (255, 0, 0)        # one outlier inside the id-7 region
    out = O.denoise(rgba, idd)
    assert np.array_equal(out[:, :30], rgba[:, :30])
    # radius 20: the outlier is one of ~1600 taps, (255-90)/255/1600 is below half an 8-bit step -> invisible
    assert np.all(out[:, 30:, :3] == 90)
    # radius law through a single differing id: distance 40000 -> radius 1
    idd2 = idd.copy()
    idd2[:, 30:, 1] = 40000
    out2 = O.denoise(rgba, idd2)
    changed2 = np.argwhere(np.any(out2[:, 30:, :3] != 90, axis=-1))
    assert np.all(np.abs(changed2[:, 0] - 20) <= 1) and np.all(np.abs(changed2[:, 1] + 30 - 45) <= 1) and len(changed2) == 9
    # mean of the 3x3 window around the outlier: (255 + 8*90)/9 etc.
    c = out2[20, 45, :3]
    assert abs(int(c[0]) - round((255 + 8 * 90) / 9)) <= 1 and abs(int(c[1]) - round(8 * 90 / 9)) <= 1


def test_byte_to_unorm_without_division_is_exact():
    """The kernels turn a byte b into (float)b / 255.0f as q = b * RN(1/255); q + fma(-q, 255, b) * RN(1/255), both steps
    fused (vrt_common.hip.h unorm_of). Exact rational arithmetic: that equals the correctly rounded quotient -- what the
    shader's division (comp:173-177) and the oracle compute -- for every byte, while the bare product does not."""
    from fractions import Fraction

    def rn(x):  # nearest float32 to the rational x, ties to even
        c = np.float32(float(x))
        cands = [np.nextafter(c, np.float32(-np.inf)), c, np.nextafter(c, np.float32(np.inf))]
        return min(cands, key=lambda v: (abs(Fraction(float(v)) - x), int(np.array(v).view(np.uint32)) & 1))

    rcp = rn(Fraction(1, 255))
    assert int(np.array(rcp).view(np.uint32)) == 0x3b808081
    bare_wrong = 0
    for b in range(256):
        want = np.float32(b) / np.float32(255.0)
        assert rn(Fraction(b, 255)) == want
        q = rn(Fraction(b) * Fraction(float(rcp)))
        bare_wrong += q != want
        e = rn(Fraction(b) - Fraction(float(q)) * 255)
        assert rn(Fraction(float(e)) * Fraction(float(rcp)) + Fraction(float(q))) == want, b
    assert bare_wrong > 100
