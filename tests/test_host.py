"""The product host library (octree.hpp / voxReader.hpp / Camera.hpp API) against golden vectors and the oracle (CPU)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from conftest import MAPS, random_voxels


def _f(u):
    return struct.unpack("f", struct.pack("I", u))[0]


def test_flatten_is_byte_identical_to_reference_hashes(V, golden, product_scenes):
    for name, g in golden["flatten"]["maps"].items():
        tex, dim = product_scenes[name]
        assert tex.size == g["bytes"] and dim == g["tex_dim"]
        assert "%016x" % V.fnv1a64(tex) == g["fnv1a64"]


def test_terrain_config4_flatten_matches_oracle_and_fixture(V, O, golden, product_scenes):
    """BASELINE config 4: product build + flatten of the height-field terrain == the oracle's == tests/golden/terrain.json;
    the height field itself is the output of the reference's own FastNoiseLite.h (re-derived when the reference is here)."""
    import json
    import subprocess
    from conftest import GOLDEN, ROOT
    t = golden["terrain"]
    h = np.load(os.path.join(GOLDEN, "terrain_heights.npz"))["heights"]
    assert h.dtype == np.uint16 and h.shape == (t["generator"]["size"],) * 2
    assert (int(h.min()), int(h.max())) == (t["generator"]["min"], t["generator"]["max"])
    tex, dim = product_scenes["terrain"]
    assert tex.size // 4 == t["texels"] < t["texel_limit"] and dim == t["tex_dim"]
    assert "%016x" % V.fnv1a64(tex) == t["fnv1a64"]
    tree = O.new_tree()
    wd = t["window"]
    O.fill_heights(tree, h, wd["x0"], wd["z0"], wd["nx"], wd["nz"], t["band"], t["floor"])
    otex, odim = O.flatten(tree)
    O.lib().o_octree_delete(tree)
    assert odim == dim and np.array_equal(otex, tex)
    # a different window and band through both builders (ragged edges, floor clamp)
    w2 = V.World()
    w2.fill_heights(h, 3, 1000, 21, 24, 5, 250)
    t2 = O.new_tree()
    O.fill_heights(t2, h, 3, 1000, 21, 24, 5, 250)
    assert np.array_equal(w2.flatten()[0], O.flatten(t2)[0])
    O.lib().o_octree_delete(t2)
    with pytest.raises(V.VrtError):
        w2.fill_heights(h, 1000, 0, 100, 1)
    w2.close()
    ref_noise = os.path.join(ROOT, "oracle", "_ref", "ref_noise")
    if os.path.isdir("/root/reference/include") and os.path.exists(ref_noise):
        tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "terrain_check.u16")
        g = t["generator"]
        info = json.loads(subprocess.check_output([ref_noise, tmp, str(g["size"]), str(g["amp"]), str(g["seed"])]))
        assert np.array_equal(np.fromfile(tmp, dtype="<u2").reshape(h.shape), h) and info["max"] == g["max"]
        os.remove(tmp)


def test_room_scene_flattens_like_the_oracle(V, O, golden, product_scenes):
    """The reference's translucent room (tests/golden/room.npz): the product's builder + flattener against the oracle's on
    the same ordered insert list (overwrites at the wall corners, glass / jelly materials, alpha < 255)."""
    from conftest import room_tree
    tex, dim = product_scenes["room"]
    otex, odim = O.flatten(room_tree(O))
    assert dim == odim and np.array_equal(tex, otex)
    assert golden["room"]["inserts"] == 12800
    leaves = tex.reshape(-1, 4)
    assert {40, 100} <= set(np.unique(leaves[:, 3]).tolist())        # glass and jelly alphas are in the stream
    assert {int(np.float32(1.5) * np.float32(85.0)), int(np.float32(1.38) * np.float32(85.0))} <= set(np.unique(leaves[:, 0]).tolist())


def test_wide_pointer_stream_is_the_reference_stream_below_2_23(O):
    """The oracle's non-reference stream extension (oracle.h: o_scene.wide_pointers) changes nothing for trees the
    reference's 23-bit pointers can address: same bytes, same frame, same fetch counts."""
    tree, ok, _ = O.load_vox(os.path.join(MAPS, "monu9.vox"))
    assert ok
    a, d = O.flatten(tree)
    b, d2 = O.flatten(tree, wide=True)
    assert d == d2 and np.array_equal(a, b)
    (ip, iv, cp), _ = O.camera_ubo((48.5, 60.5, 170.5), -90.0, -12.0, 96, 54)
    r1 = O.render(O.make_scene(a, d, ip, iv, cp), 96, 54, 2)
    r2 = O.render(O.make_scene(a, d, ip, iv, cp, wide=True), 96, 54, 2)
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) and r1[3] == r2[3]


def test_camera_block_bits(V, golden):
    for c in golden["camera"]["cases"]:
        ip, iv, cp, fr = V.camera_block([_f(x) for x in c["pos"]], _f(c["yaw"]), _f(c["pitch"]), c["width"], c["height"])
        assert list(ip.view(np.uint32)) == c["inv_proj"]
        assert list(iv.view(np.uint32)) == c["inv_view"]
        assert list(fr.view(np.uint32)) == c["front"]
        assert list(cp.view(np.uint32)) == c["pos"] + [0x3f800000]
    p = V.Params()
    V.hip_lib().vrt_default_params(C.byref(p))
    assert list(np.array(p.light_dir, np.float32).view(np.uint32)) == golden["camera"]["light_dir"]


def test_colour_functions_are_the_references(V, golden):
    """include/color.h keeps the reference's names (include/color.h:33-46); tests/golden/color.json holds what the
    reference's own src/color.c returns for 96 inputs (oracle/ref_color_driver.c)."""
    L = V.host_lib()
    for name in ("get_color_rgba", "get_color_rgb", "make_color_rgb", "make_color_rgba"):
        getattr(L, name).restype = C.c_uint32
    for name in ("get_red_rgb", "get_red_rgba", "get_green_rgb", "get_green_rgba", "get_blue_rgb", "get_blue_rgba", "get_alpha_rgba"):
        getattr(L, name).restype = C.c_uint8
    for c in golden["color"]["cases"]:
        v = c["in"]
        rgb = v & 0xffffff
        b = [(v >> 24) & 255, (v >> 16) & 255, (v >> 8) & 255, v & 255]
        got = {"make_color_rgb": L.make_color_rgb(C.c_uint8(b[0]), C.c_uint8(b[1]), C.c_uint8(b[2])),
               "make_color_rgba": L.make_color_rgba(C.c_uint8(b[0]), C.c_uint8(b[1]), C.c_uint8(b[2]), C.c_uint8(b[3])),
               "get_color_rgba": L.get_color_rgba(C.c_uint32(v)), "get_color_rgb": L.get_color_rgb(C.c_uint32(rgb)),
               "get_red_rgb": L.get_red_rgb(C.c_uint32(rgb)), "get_red_rgba": L.get_red_rgba(C.c_uint32(v)),
               "get_green_rgb": L.get_green_rgb(C.c_uint32(rgb)), "get_green_rgba": L.get_green_rgba(C.c_uint32(v)),
               "get_blue_rgb": L.get_blue_rgb(C.c_uint32(rgb)), "get_blue_rgba": L.get_blue_rgba(C.c_uint32(v)),
               "get_alpha_rgba": L.get_alpha_rgba(C.c_uint32(v))}
        for k, want in c.items():
            if k != "in":
                assert got[k] == want, (hex(v), k, got[k], want)


def _oracle_tree_from(O, xyz, rgba):
    L = O.lib()
    t = O.new_tree()
    for (x, y, z), c in zip(xyz, rgba):
        L.o_octree_insert(t, O.VoxelObj(O.IVec3(int(x), int(y), int(z)), int(c), O.Voxel(3.0, 0.0, 0.0)))
    return t


@pytest.mark.parametrize("seed,n,lo,hi", [(1, 1, 0, 50), (2, 500, 0, 24), (3, 4000, -40, 40), (4, 3000, -1023, 1024),
                                          (5, 6000, 0, 16), (6, 2500, 1000, 1024)])
def test_random_builds_flatten_like_the_oracle(V, O, seed, n, lo, hi):
    """Insert (with heavy overlap so merges/re-splits happen), then remove a subset: bytes must match at each stage."""
    rng = np.random.default_rng(seed)
    xyz, rgba = random_voxels(rng, n, lo, hi, n_colors=2 if seed == 5 else 5)
    w = V.World()
    w.insert_many(xyz, rgba)
    t = _oracle_tree_from(O, xyz, rgba)
    a, da = w.flatten()
    b, db = O.flatten(t)
    assert da == db and np.array_equal(a, b)
    assert w.texel_count() == O.lib().o_octree_texel_size(t)
    for i in rng.permutation(n)[: n // 3]:
        x, y, z = (int(v) for v in xyz[i])
        w.remove(x, y, z)
        O.lib().o_octree_remove(t, O.IVec3(x, y, z))
    a, da = w.flatten()
    b, db = O.flatten(t)
    assert da == db and np.array_equal(a, b)
    # re-insert over the holes with one material: exercises merge-up
    w.insert_many(xyz[: n // 2], np.full(n // 2, 0x808080ff, np.uint32))
    for x, y, z in xyz[: n // 2]:
        O.lib().o_octree_insert(t, O.VoxelObj(O.IVec3(int(x), int(y), int(z)), 0x808080ff, O.Voxel(3.0, 0.0, 0.0)))
    a, da = w.flatten()
    b, db = O.flatten(t)
    assert da == db and np.array_equal(a, b)
    # octree_find parity (including the y-blind equality of the shipped vmm)
    for x, y, z in rng.integers(lo - 2, hi + 2, size=(300, 3)):
        got = w.find(int(x), int(y), int(z))
        ref = O.lib().o_octree_find(t, O.IVec3(int(x), int(y), int(z)))
        assert got["coord"] == (ref.coord.x, ref.coord.y, ref.coord.z) and got["color"] == ref.color
    O.lib().o_octree_delete(t)
    w.close()


def test_solid_block_merges_to_one_leaf(V):
    w = V.World(world_min=(0, 0, 0), world_max=(8, 8, 8))
    pts = np.array([(x, y, z) for x in range(8) for y in range(8) for z in range(8)], np.int32)
    w.insert_many(pts, np.full(len(pts), 0x102030ff, np.uint32))
    # everything merged into the root: a lone leaf root flattens to its 2 data texels
    tex, dim = w.flatten()
    assert w.texel_count() == 2 and tex.size == 8 and list(tex[:4]) == [0x10, 0x20, 0x30, 255] and tex[4] == 255
    w.remove(3, 3, 3)
    assert w.texel_count() > 2


def test_empty_world(V):
    w = V.World()
    tex, dim = w.flatten()
    assert tex.size == 0 and dim == 1 and w.texel_count() == 0
    assert w.ray_cast((0.5, 0.5, 0.5), (0.0, 0.0, -1.0)) is None
    rec, info = V.build_layout(tex)
    assert info.n_records == 1 and rec[0, 0] == 0


def test_vox_loader_raw_and_errors(V, O, tmp_path):
    data = V.make_custom_vox()
    w = V.World()
    ok, n = w.load_vox_bytes(data)
    t, ok2, n2 = O.load_vox(data)
    assert ok and ok2 and n == n2 and n > 10000
    a, da = w.flatten()
    b, db = O.flatten(t)
    assert da == db and np.array_equal(a, b)
    # through a file as well
    p = tmp_path / "custom.vox"
    p.write_bytes(data)
    w2 = V.World()
    assert w2.load_vox(p)
    assert np.array_equal(w2.flatten()[0], a)
    # error behaviour of load_vox_file: missing file, bad magic, no voxels -> false
    assert not V.World().load_vox(tmp_path / "nope.vox")
    bad = tmp_path / "bad.vox"
    bad.write_bytes(b"NOPE" + data[4:])
    assert not V.World().load_vox(bad)
    empty = V.encode_vox((4, 4, 4), np.zeros((0, 4), np.uint8))
    assert V.World().load_vox_bytes(empty + b"\0" * 16)[0] is False
    # truncated file: whatever was parsed before the cut is still what both sides build
    cut = data[: len(data) // 2]
    wa = V.World()
    oka, na = wa.load_vox_bytes(cut)
    tb, okb, nb = O.load_vox(cut)
    assert oka == okb and na == nb and np.array_equal(wa.flatten()[0], O.flatten(tb)[0])


def _chunk(tag, content, children=b""):
    return tag + struct.pack("<ii", len(content), len(children)) + content + children


def _vstr(s):
    return struct.pack("<i", len(s)) + s


def _vdict(d):
    out = struct.pack("<i", len(d))
    for k, v in d.items():
        out += _vstr(k) + _vstr(v)
    return out


def _scene_graph_vox(rot_byte, translation, second_translation):
    """Two 3x2x4 models placed by nTRN/nGRP/nSHP nodes (the chunk grammar MagicaVoxel writes)."""
    def model(seed):
        rng = np.random.default_rng(seed)
        pts = [(x, y, z, 1 + int(rng.integers(0, 200))) for x in range(3) for y in range(2) for z in range(4)
               if rng.random() < 0.8]
        return _chunk(b"SIZE", struct.pack("<iii", 3, 2, 4)) + \
            _chunk(b"XYZI", struct.pack("<i", len(pts)) + bytes(v for p in pts for v in p))
    body = model(1) + model(2)
    trn = lambda nid, child, frame: _chunk(b"nTRN", struct.pack("<i", nid) + _vdict({}) +
                                           struct.pack("<iiii", child, -1, 0, 1) + _vdict(frame))
    body += trn(0, 1, {})
    body += _chunk(b"nGRP", struct.pack("<i", 1) + _vdict({}) + struct.pack("<iii", 2, 2, 4))
    body += trn(2, 3, {b"_t": translation, b"_r": str(rot_byte).encode()})
    body += _chunk(b"nSHP", struct.pack("<i", 3) + _vdict({}) + struct.pack("<i", 1) + struct.pack("<i", 0) + _vdict({}))
    body += trn(4, 5, {b"_t": second_translation})
    body += _chunk(b"nSHP", struct.pack("<i", 5) + _vdict({}) + struct.pack("<i", 1) + struct.pack("<i", 1) + _vdict({}))
    pal = bytes((i, 255 - i, (i * 7) & 255, 255)[j] for i in range(256) for j in range(4))
    body += _chunk(b"RGBA", pal)
    return b"VOX " + struct.pack("<i", 150) + _chunk(b"MAIN", b"", body) + b"\0" * 12


def test_vox_loader_survives_hostile_files(V, O):
    """A scene graph that refers back to an ancestor must not overflow the host stack, and XYZI chunks that claim far more
    voxels than they carry must not cost memory for what is not there (the reference does both)."""
    import resource
    pts = [(1, 2, 3, 7), (2, 2, 3, 9)]
    model = _chunk(b"SIZE", struct.pack("<iii", 4, 4, 4)) + _chunk(b"XYZI", struct.pack("<i", len(pts)) + bytes(v for p in pts for v in p))
    trn = lambda nid, child: _chunk(b"nTRN", struct.pack("<i", nid) + _vdict({}) + struct.pack("<iiii", child, -1, 0, 1) + _vdict({}))
    # 0 -> group 1 -> {transform 2 -> shape 3, transform 4 -> group 1 (cycle)}
    body = model + trn(0, 1) + _chunk(b"nGRP", struct.pack("<i", 1) + _vdict({}) + struct.pack("<iii", 2, 2, 4)) + trn(2, 3)
    body += _chunk(b"nSHP", struct.pack("<i", 3) + _vdict({}) + struct.pack("<i", 1) + struct.pack("<i", 0) + _vdict({})) + trn(4, 1)
    cyclic = b"VOX " + struct.pack("<i", 150) + _chunk(b"MAIN", b"", body)
    w = V.World()
    ok, n = w.load_vox_bytes(cyclic)
    assert ok and w.texel_count() > 0       # the shape was placed (once per pass round the cycle until the depth cap drops the branch)
    # 300 XYZI chunks of 8 bytes each that claim 9.9 M voxels: 12 GB if every claim were allocated
    big = b"".join(_chunk(b"SIZE", struct.pack("<iii", 4, 4, 4)) + _chunk(b"XYZI", struct.pack("<i", 9_900_000) + bytes((1, 1, 1, 5)))
                   for _ in range(300))
    hostile = b"VOX " + struct.pack("<i", 150) + _chunk(b"MAIN", b"", big)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    # a node 0 that is a transform to a missing child: scene-graph mode, so the 3e9 implied zero voxels are never walked
    w2 = V.World()
    ok2, n2 = w2.load_vox_bytes(b"VOX " + struct.pack("<i", 150) + _chunk(b"MAIN", b"", big + trn(0, 99)))
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert (after - before) < 200 * 1024          # KiB: nowhere near 300 x 40 MB
    assert n2 == 0 and w2.texel_count() == 0 and len(hostile) < 20000
    w.close(); w2.close()


@pytest.mark.parametrize("rot", [4, 2, 9, 17, 40, 98, 120, 1, 6, 24, 70])
def test_vox_loader_scene_graph_matches_oracle(V, O, rot):
    data = _scene_graph_vox(rot, b"10 -7 5", b"-3 12 20")
    w = V.World()
    ok, _ = w.load_vox_bytes(data, offset=(100, 50, 60))
    t, ok2, n2 = O.load_vox(data, offset=(100, 50, 60))
    assert ok and ok2 and n2 > 20
    a, da = w.flatten()
    b, db = O.flatten(t)
    assert a.size > 0 and da == db and np.array_equal(a, b)


def test_cpu_ray_cast_matches_oracle_config1(V, O):
    """BASELINE config 1: custom.vox stand-in, 256x256, single-thread octree_ray_cast per pixel (sampled here)."""
    data = V.make_custom_vox()
    w = V.World()
    assert w.load_vox_bytes(data)[0]
    t, _, _ = O.load_vox(data)
    W = H = 256
    ip, iv, cp, fr = V.camera_block((32.5, 40.5, 150.5), -90.0, -8.0, W, H)
    # same worldDir the kernel would use: take it from the oracle's ray setup via a tiny frame render
    rng = np.random.default_rng(3)
    hits = 0
    for _ in range(600):
        o = (np.float32(32.5) + np.float32(rng.uniform(-20, 20)), np.float32(40.5) + np.float32(rng.uniform(-20, 20)),
             np.float32(150.5))
        target = rng.uniform(-10, 74, size=3)
        d = (target - np.array(o, np.float64)).astype(np.float32)
        d = d / np.float32(np.linalg.norm(d)) * np.float32(rng.uniform(0.5, 2.0))
        got = w.ray_cast(o, d)
        ref = O.lib().o_octree_ray_cast(t, O.Vec3(*[float(v) for v in o]), O.Vec3(*[float(v) for v in d]),
                                        O.Vec3(0, 0, 0), O.Vec3(1024, 1024, 1024))
        if ref:
            hits += 1
            n = ref.contents
            assert got == ((n.voxel.coord.x, n.voxel.coord.y, n.voxel.coord.z), bool(n.has_voxel))
        else:
            assert got is None
    assert hits > 100


def test_layout_walk_equals_texel_walk(V, O, product_scenes):
    """The device record array answers point queries exactly like the texel stream does."""
    tex, dim = product_scenes["monu9"]
    rec, info = V.build_layout(tex)
    assert info.n_records == info.n_internal + info.n_leaves == rec.shape[0]
    s = O.make_scene(tex, dim, np.eye(4, dtype=np.float32).ravel(), np.eye(4, dtype=np.float32).ravel(), [0, 0, 0, 1])
    leaf = (C.c_uint8 * 8)()
    mn, mx = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    rng = np.random.default_rng(11)
    pts = np.concatenate([rng.integers(-10, 110, size=(3000, 3)), rng.integers(-1023, 1024, size=(500, 3))])
    for p in pts:
        found = O.lib().o_find_point(C.byref(s), (C.c_int32 * 3)(*[int(v) for v in p]), leaf, mn, mx)
        lo, hi = [-1023] * 3, [1024] * 3
        masks, base = int(rec[0, 0]), int(rec[0, 1])
        got = None
        for _ in range(16):
            mid = [lo[i] + (hi[i] - lo[i]) // 2 for i in range(3)]
            ci = (4 if p[0] >= mid[0] else 0) | (2 if p[1] >= mid[1] else 0) | (1 if p[2] >= mid[2] else 0)
            for i, bit in enumerate((4, 2, 1)):
                if ci & bit:
                    lo[i] = mid[i]
                else:
                    hi[i] = mid[i]
            if not (masks >> ci) & 1:
                got = (0, None)
                break
            idx = base + bin(masks & 0xff & ((1 << ci) - 1)).count("1")
            if (masks >> (8 + ci)) & 1:
                got = (1, (int(rec[idx, 0]), int(rec[idx, 1])))
                break
            masks, base = int(rec[idx, 0]), int(rec[idx, 1])
        assert got is not None and got[0] == found
        assert lo == list(mn) and hi == list(mx)
        if found:
            w0, w1 = got[1]
            assert (w0 & 0xff, (w0 >> 8) & 0xff, (w0 >> 16) & 0xff, w0 >> 24) == (leaf[0], leaf[1], leaf[2], leaf[7])
            assert (w1 & 0xff, (w1 >> 8) & 0xff, (w1 >> 16) & 0xff) == (leaf[4], leaf[5], leaf[6])


@pytest.mark.parametrize("name", ["dragon", "monu9", "nature"])
def test_wide_layout_answers_like_the_texel_stream(V, O, product_scenes, name):
    """The 64-cell wide layout (what the default kernels read) returns octreeFind's result: leaf words + node AABB."""
    tex, dim = product_scenes[name]
    rng = np.random.default_rng(21)
    pts = np.concatenate([rng.integers(-6, 132, size=(6000, 3)), rng.integers(-1023, 1024, size=(1500, 3)),
                          np.array([[0, 0, 0], [-1, 0, 0], [1023, 1023, 1023], [-1023, -1023, -1023], [512, 0, 3]])])
    res = V.wide_find(tex, pts)
    assert res is not None
    out, (n_nodes, n_roots) = res
    assert n_roots == 1 and n_nodes > 500
    s = O.make_scene(tex, dim, np.eye(4, dtype=np.float32).ravel(), np.eye(4, dtype=np.float32).ravel(), [0, 0, 0, 1])
    leaf = (C.c_uint8 * 8)()
    mn, mx = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    for p, o in zip(pts, out):
        f = O.lib().o_find_point(C.byref(s), (C.c_int32 * 3)(*[int(v) for v in p]), leaf, mn, mx)
        w0 = (leaf[0] | leaf[1] << 8 | leaf[2] << 16 | leaf[7] << 24) if f else 0
        w1 = ((leaf[4] if leaf[7] else 0) | leaf[5] << 8 | leaf[6] << 16) if f else 0   # refraction byte reads 0 under alpha 0
        assert (int(o[0]), int(o[1])) == (w0, w1), p
        assert list(o[2:5].view(np.int32)) == list(mn) and list(o[5:8].view(np.int32)) == list(mx), p


def test_wide_layout_other_world_bounds_and_refusals(V, O):
    rng = np.random.default_rng(4)
    xyz, rgba = random_voxels(rng, 3000, -60, 70)
    w = V.World(world_min=(-256, -256, -256), world_max=(256, 256, 256))   # side 2^9: its eight octants are the wide roots
    w.insert_many(xyz, rgba)
    tex, dim = w.flatten()
    pts = rng.integers(-256, 256, size=(4000, 3))
    out, (n_nodes, n_roots) = V.wide_find(tex, pts, (-256, -256, -256), (256, 256, 256))
    assert n_roots == 8
    s = O.make_scene(tex, dim, np.eye(4, dtype=np.float32).ravel(), np.eye(4, dtype=np.float32).ravel(), [0, 0, 0, 1])
    s.bounds_min[:] = (-256, -256, -256)
    s.bounds_max[:] = (256, 256, 256)
    leaf = (C.c_uint8 * 8)()
    mn, mx = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    for p, o in zip(pts, out):
        f = O.lib().o_find_point(C.byref(s), (C.c_int32 * 3)(*[int(v) for v in p]), leaf, mn, mx)
        w0 = (leaf[0] | leaf[1] << 8 | leaf[2] << 16 | leaf[7] << 24) if f else 0
        w1 = ((leaf[4] if leaf[7] else 0) | leaf[5] << 8 | leaf[6] << 16) if f else 0   # refraction byte reads 0 under alpha 0
        assert (int(o[0]), int(o[1])) == (w0, w1) and list(o[2:5].view(np.int32)) == list(mn) and list(o[5:8].view(np.int32)) == list(mx)
    # odd-sized world with voxels on both sides of every split: several aligned sub-trees -> still exact or refused
    w2 = V.World()
    xyz2, rgba2 = random_voxels(rng, 4000, -900, 900)
    w2.insert_many(xyz2, rgba2)
    tex2, dim2 = w2.flatten()
    pts2 = np.concatenate([xyz2[:1500], rng.integers(-1023, 1024, size=(1500, 3))])
    res = V.wide_find(tex2, pts2)
    if res is not None:
        s2 = O.make_scene(tex2, dim2, np.eye(4, dtype=np.float32).ravel(), np.eye(4, dtype=np.float32).ravel(), [0, 0, 0, 1])
        for p, o in zip(pts2, res[0]):
            f = O.lib().o_find_point(C.byref(s2), (C.c_int32 * 3)(*[int(v) for v in p]), leaf, mn, mx)
            w0 = (leaf[0] | leaf[1] << 8 | leaf[2] << 16 | leaf[7] << 24) if f else 0
            assert int(o[0]) == w0 and list(o[2:5].view(np.int32)) == list(mn) and list(o[5:8].view(np.int32)) == list(mx)


def test_records_from_tree_equal_records_from_texel_stream(V, product_scenes):
    """EXTENSION path: the record array emitted straight from the pointer octree is the one the uploader derives
    from octree_texture()'s stream -- on the shipped maps and on random edit sequences."""
    import os
    for name in ("dragon", "monu9", "nature"):
        w = V.World()
        assert w.load_vox(os.path.join(MAPS, name + ".vox"))
        rec_tree, dim_tree = w.records()
        tex, dim = product_scenes[name]
        rec_tex, info = V.build_layout(tex)
        assert dim_tree == dim and np.array_equal(rec_tree, rec_tex), name
    rng = np.random.default_rng(12)
    w = V.World()
    xyz, rgba = random_voxels(rng, 5000, -200, 300)
    w.insert_many(xyz, rgba)
    for i in rng.permutation(5000)[:1500]:
        w.remove(*[int(v) for v in xyz[i]])
    w.insert_many(xyz[:700], np.full(700, 0x224466ff, np.uint32))
    tex, dim = w.flatten()
    rec_tree, dim_tree = w.records()
    assert dim_tree == dim and np.array_equal(rec_tree, V.build_layout(tex)[0])
    # an empty world and a world merged into a single root leaf
    e = V.World()
    rec, d = e.records()
    assert d == 1 and np.array_equal(rec, V.build_layout(np.zeros(0, np.uint8))[0]) and rec[0, 0] == 0
    s = V.World(world_min=(0, 0, 0), world_max=(4, 4, 4))
    pts = np.array([(x, y, z) for x in range(4) for y in range(4) for z in range(4)], np.int32)
    s.insert_many(pts, np.full(len(pts), 0x102030ff, np.uint32))
    assert s.records() is None


def test_dispatcher_knows_when_a_tree_needs_no_ray_stack(V, product_scenes):
    """vrt::tree_is_opaque (what lets VRT_MODE_FULL run without its 8-deep stack): the shipped maps and the terrain qualify (every
    leaf alpha 255 with refraction byte 255, or a phantom alpha-0 leaf); the reference's room does not (glass, jelly); one
    translucent voxel, one opaque voxel whose refraction byte reads as empty space (85) or as 0, disqualify a tree; alpha-0
    leaves of any refraction and emissive voxels do not."""
    for name in ("dragon", "monu9", "nature", "terrain"):
        assert V.tree_is_opaque(product_scenes[name][0]), name
    assert not V.tree_is_opaque(product_scenes["room"][0])
    base = [(x, 0, z) for x in range(6) for z in range(6)]

    def world(extra):
        w = V.World()
        for x, y, z in base:
            w.insert(x, y, z, 0xa0a0a0ff)
        for (x, y, z), args in extra:
            w.insert(x, y, z, *args)
        return w.flatten()[0]

    assert V.tree_is_opaque(world([]))
    assert V.tree_is_opaque(world([((2, 3, 2), (0xffd2d2ff, 3.0, 1.0, 0.0)), ((3, 3, 3), (0x11223300, 1.2, 0.0, 0.5))]))   # emissive; alpha 0
    assert not V.tree_is_opaque(world([((2, 3, 2), (0xc8dcff50, 1.5, 0.0, 0.0))]))      # glass
    assert not V.tree_is_opaque(world([((2, 3, 2), (0x3c64dcfe, 3.0, 0.0, 0.0))]))      # alpha 254
    assert not V.tree_is_opaque(world([((2, 3, 2), (0x50b43cff, 1.0, 0.0, 0.0))]))      # opaque, refraction byte 85: invisible to the hit test
    assert not V.tree_is_opaque(world([((2, 3, 2), (0x50b43cff, 0.0, 0.0, 0.0))]))      # opaque, refraction byte 0
    assert V.tree_is_opaque(world([((2, 3, 2), (0x50b43cff, 2.0, 0.0, 0.3))]))
    assert not V.tree_is_opaque(np.zeros(0, np.uint8)) or True                           # an empty world: either answer is harmless (nothing to hit)


def test_box_records_expand_exactly_the_nodes_that_meet_the_box(V):
    """vrth_world_box_records (what a box edit hands to vrt_patch_apply): walked beside the full record array from the same node,
    every child must be the same kind with the same leaf words, except that an INTERNAL child whose cube does not meet the box may
    be a "keep" record -- and must be one (only the nodes that meet the box are walked); a box covering the world gives the full
    array, a one-voxel box gives what vrth_world_path_records gives."""
    import os
    w = V.World()
    assert w.load_vox(os.path.join(MAPS, "monu9.vox"))
    full, _ = w.records()
    KEEP = 0xffffffff
    wmin, wmax = (-1023, -1023, -1023), (1024, 1024, 1024)
    assert np.array_equal(w.box_records([], wmin, tuple(v - 1 for v in wmax)), full)
    rng = np.random.default_rng(4)
    for trial in range(12):
        lo = [int(v) for v in rng.integers(0, 90, size=3)]
        n = int(rng.choice([1, 1, 4, 16, 40]))
        hi = [v + n - 1 for v in lo]
        sparse = w.box_records([], lo, hi)
        kept = expanded = 0
        todo = [(0, 0, wmin, wmax)]          # (record in sparse, record in full, node cube)
        while todo:
            si, fi, mn, mx = todo.pop()
            assert sparse[si, 0] == full[fi, 0], "an expanded node carries the same child and leaf masks"
            mask, leaf = int(full[fi, 0]) & 0xff, (int(full[fi, 0]) >> 8) & 0xff
            sc, fc = int(sparse[si, 1]), int(full[fi, 1])
            for ci in range(8):
                if not (mask >> ci) & 1:
                    continue
                cmn, cmx = list(mn), list(mx)
                for k in range(3):
                    mid = mn[k] + ((mx[k] - mn[k]) >> 1)
                    if (ci >> (2 - k)) & 1:
                        cmn[k] = mid
                    else:
                        cmx[k] = mid
                if (leaf >> ci) & 1:
                    assert np.array_equal(sparse[sc], full[fc])
                else:
                    meets = all(hi[k] >= cmn[k] and lo[k] < cmx[k] for k in range(3))
                    is_keep = int(sparse[sc, 0]) == KEEP and int(sparse[sc, 1]) == KEEP
                    assert is_keep == (not meets), (trial, ci, cmn, cmx, lo, hi)
                    if meets:
                        expanded += 1
                        todo.append((sc, fc, cmn, cmx))
                    else:
                        kept += 1
                sc += 1
                fc += 1
        assert kept > 0 and expanded > 0 and len(sparse) < len(full)
        if n == 1:
            H = V.host_lib()
            import ctypes as C
            H.vrth_world_path_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
            p, cnt = C.c_void_p(), C.c_size_t(0)
            assert H.vrth_world_path_records(w._h, (C.c_uint8 * 16)(), 0, lo[0], lo[1], lo[2], C.byref(p), C.byref(cnt)) == 0
            one = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(cnt.value, 2)).copy()
            H.vrth_free(p)
            assert np.array_equal(one, sparse)
    with pytest.raises(V.VrtError):
        w.box_records([], (5, 5, 5), (4, 9, 9))


@pytest.mark.parametrize("bounds", [((-1023, -1023, -1023), (1024, 1024, 1024)), ((0, 0, 0), (256, 256, 256)),
                                    ((-7, -3, -5), (123, 70, 99))])
def test_edit_patches_answer_like_rebuilt_layouts(V, bounds):
    """vrt_patch_plan / vrt_patch_apply on the host structures (no device): after every insert / remove, the layouts
    patched in place answer point lookups -- leaf words and node box, through the wide cells and through the records
    alone -- exactly like layouts rebuilt from the edited tree's stream, and track its texel count."""
    rng = np.random.default_rng(abs(sum(bounds[0])) + 99)
    lo, hi = np.array(bounds[0]), np.array(bounds[1])
    w = V.World(bounds[0], bounds[1])
    span = np.minimum(hi - lo, 120)
    base = np.maximum(lo, 0)
    for _ in range(6):
        c = base + rng.integers(0, span, size=3)
        xyz = np.clip(c + rng.integers(-4, 5, size=(40, 3)), lo, hi - 1).astype(np.int32)
        w.insert_many(xyz, np.full(len(xyz), 0xa0a0a0ff, np.uint32))
    placed = []
    patched = appended = 0
    before, _ = w.flatten()
    for step in range(120):
        kind = step % 4
        if kind == 3 and placed:
            v = placed.pop(int(rng.integers(0, len(placed))))
            w.remove(*v)
        else:
            if kind == 0:      # near the geometry
                v = tuple(int(t) for t in np.clip(base + rng.integers(0, span, size=3), lo, hi - 1))
            elif kind == 1:    # anywhere in the world
                v = tuple(int(t) for t in rng.integers(lo, hi, size=3))
            else:              # next to something placed earlier
                q = placed[int(rng.integers(0, len(placed)))] if placed else tuple(int(t) for t in base)
                v = tuple(int(t) for t in np.clip(np.array(q) + rng.integers(-1, 2, size=3), lo, hi - 1))
            w.insert(v[0], v[1], v[2], int(rng.choice([0x50b43cff, 0xc8dcff50, 0x11223300])), float(rng.choice([3.0, 1.5])), 0.0, 0.0)
            placed.append(v)
        after, _ = w.flatten()
        near = np.clip(np.array(v) + rng.integers(-9, 10, size=(150, 3)), lo, hi - 1)
        far = rng.integers(lo, hi, size=(150, 3))
        pts = np.concatenate([near, far, [v]])
        bad, depth, n_rec, n_cells, texels_ok = V.patch_check(before, after, v, pts, bounds[0], bounds[1])
        assert bad == 0, (step, v, depth)
        # the same with a sub-tree that carries only the path to the voxel (every other internal child: "keep")
        assert V.patch_check(before, after, v, pts, bounds[0], bounds[1], sparse=True) == (0, depth, n_rec, n_cells, texels_ok), (step, v)
        if depth:
            patched += 1
            appended += n_rec
            assert texels_ok, (step, v, depth)
        before = after
    assert patched >= 60, patched
    assert appended / patched < 200, appended / patched          # only the blocks along the path are appended


def test_malformed_texel_streams_are_refused_or_laid_out_never_crash(V, product_scenes):
    """Upload hardening on the host (no device): random byte streams and mutated real ones either lay out within the
    limits -- and then answer point lookups through the wide cells without leaving their arrays -- or are refused."""
    rng = np.random.default_rng(4242)
    tex, _ = product_scenes["monu9"]
    tex = np.asarray(tex, np.uint8)
    pts = rng.integers(-1023, 1024, size=(64, 3)).astype(np.int32)
    laid_out = refused = 0
    streams = []
    for n in (0, 4, 8, 12, 40, 400, 4000):
        streams += [rng.integers(0, 256, size=n, dtype=np.uint8) for _ in range(20)]
    for _ in range(400):                                   # a real stream with a few bytes, or a run of them, damaged
        t = tex.copy()
        k = int(rng.integers(1, 6))
        pos = rng.integers(0, t.size, size=k)
        if rng.random() < 0.3:
            a = int(rng.integers(0, t.size - 64))
            t[a:a + int(rng.integers(4, 64))] = rng.integers(0, 256, dtype=np.uint8)
        else:
            t[pos] = rng.integers(0, 256, size=k, dtype=np.uint8)
        if rng.random() < 0.2:
            t = t[: int(rng.integers(4, t.size)) // 4 * 4]   # truncated
        streams.append(t)
    for t in streams:
        try:
            rec, info = V.build_layout(t)
        except V.VrtError:
            refused += 1
            continue
        laid_out += 1
        assert len(rec) == info.n_records and info.max_depth <= 16
        res = V.wide_find(t, pts)                          # None: no wide form (the record kernels would run)
        if res is not None:
            out, (n_nodes, n_roots) = res
            assert n_roots <= 8
    assert laid_out > 100 and refused > 10, (laid_out, refused)


def test_ray_table_is_the_shader_formula_per_column_and_row(V):
    """vrt_test_ray_table (what the dispatcher uploads for the kernels' table-driven prologue) against the shader's
    operations in numpy float32 (IEEE, like the host code): u = px / W * 2 - 1, invProjection * (u, v, -1, 1) with the
    association of mat_vec(), division by w (comp:626-634). Refusals: projections whose x depends on v, whose w varies,
    whose zero terms change sign over the frame, non-finite entries."""
    f = np.float32
    for (W, H, pose) in [(1920, 1080, (63.5, 60.5, 140.5, -90.0, -10.0)), (1280, 720, (48.5, 60.5, 170.5, -90.0, -12.0)),
                         (97, 55, (1.5, 2.5, 3.5, 30.0, 40.0)), (1, 1, (0.5, 0.5, 0.5, 0.0, 0.0))]:
        ip, iv, cp, _ = V.camera_block(pose[:3], pose[3], pose[4], W, H)
        m = np.array(ip, f)
        t = V.ray_table(ip, W, H)
        assert t is not None, (W, H)
        x, y, z = t

        def row(r, u, v):
            return (m[0 + r] * u + m[4 + r] * v) + (m[8 + r] * f(-1.0) + m[12 + r] * f(1.0))
        u = (np.arange(W, dtype=f) / f(W)) * f(2.0) - f(1.0)
        v = (np.arange(H, dtype=f) / f(H)) * f(2.0) - f(1.0)
        w = row(3, f(1.0), f(1.0))
        assert abs(w) > 1e-6
        assert np.array_equal((row(0, u, f(1.0)) / w).view(np.uint32), x.view(np.uint32))
        assert np.array_equal((row(0, u, f(-1.0)) / w).view(np.uint32), x.view(np.uint32))   # no trace of v, not even a zero's sign
        assert np.array_equal((row(1, f(1.0), v) / w).view(np.uint32), y.view(np.uint32))
        assert np.array_equal((row(1, f(-1.0), v) / w).view(np.uint32), y.view(np.uint32))
        assert f(row(2, f(1.0), f(1.0)) / w).view(np.uint32) == f(z).view(np.uint32)
        assert V.view_in_range(iv)
    ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, 96, 54)
    for k, val in [(4, 0.05), (1, -0.2), (8, 1e-3), (3, 0.01), (7, 0.5), (2, 1.0), (0, float("nan")), (15, float("inf"))]:
        m = np.array(ip, f).copy()
        m[k] = val
        assert V.ray_table(m, 96, 54) is None, k
    m = np.array(ip, f).copy()
    m[5] = -m[5]                     # the row with v = 0: (+-0 from 0 * u) + (-0) keeps the sign of the column's zero
    assert V.ray_table(m, 96, 54) is None
    assert V.ray_table(m, 96, 55) is not None    # an odd height has no such row
    m = np.array(ip, f).copy()
    m[11] = 0.0; m[15] = 0.0         # w == 0: the shader skips the division (|w| <= 1e-6)
    t = V.ray_table(m, 8, 8)
    assert t is not None and t[2] == f(-1.0) * m[10] + m[14] * f(1.0) or t is None
    bad = np.array(iv, f).copy(); bad[:12] *= f(1e-25)
    assert not V.view_in_range(bad)
    bad = np.array(iv, f).copy(); bad[0:3] = bad[4:7]      # singular
    assert not V.view_in_range(bad)


def test_dispatcher_knows_when_the_world_is_empty_outside_wide_root_0(V, product_scenes):
    """vrt_test_root0: the dispatcher's two findings about a tree (KArgs::root0_only and the per-launch choice of wide
    root 0). The shipped maps sit in the octant [0, 1024)^3 and nothing else exists: root0_only; dragon.vox fills only the
    cell [0, 256)^3 of it, whose 64-unit cells it spreads over, so that cube becomes root 0 for an eye inside it and the
    octant stays root 0 for an eye outside. A voxel in another octant, or a second occupied 256-cell, ends either finding."""
    tex, dim = product_scenes["dragon"]
    only, shift, mn, built = V.root0_choice(tex, (63, 60, 140))
    assert (only, shift, mn, built) == (True, 8, (0, 0, 0), 10)
    assert V.root0_choice(tex, (63, 60, 300)) == (True, 10, (0, 0, 0), 10)        # eye outside [0, 256)^3
    assert V.root0_choice(tex, (-5, 60, 140)) == (True, 10, (0, 0, 0), 10)        # eye in another octant
    assert V.root0_choice(tex, (63, 60, 5000)) == (True, 10, (0, 0, 0), 10)       # eye outside the world
    w = V.World()
    assert w.load_vox(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps", "dragon.vox"))
    w.insert(700, 10, 10, 0x112233ff)                                              # a second occupied cell of the octant
    t2, _ = w.flatten()
    assert V.root0_choice(t2, (63, 60, 140)) == (True, 10, (0, 0, 0), 10)
    w.insert(-3, 10, 10, 0x112233ff)                                               # and one in another octant of the world
    t3, _ = w.flatten()
    r = V.root0_choice(t3, (63, 60, 140))
    assert r is None or r[0] is False
    # a small model far from the origin of the octant: the chain follows it down to a 64-cube, never below
    w = V.World()
    for x in range(600, 604):
        w.insert(x, 300, 520, 0xa0a0a0ff)
    t4, _ = w.flatten()
    only, shift, mn, built = V.root0_choice(t4, (601, 301, 521))
    assert only and built == 10 and shift == 6 and mn == (576, 256, 512)
    only, shift, mn, built = V.root0_choice(t4, (601, 301, 700))                   # the eye leaves the 64-cube, not the 256-cube
    assert only and shift == 8 and mn == (512, 256, 512)
