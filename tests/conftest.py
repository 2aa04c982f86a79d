import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MAPS = os.path.join(GOLDEN, "maps")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def V():
    """The product package (ctypes over libvrt_host.so / libvrt_hip.so)."""
    import vrt_import
    mod = vrt_import.vrt()
    if not (os.path.exists(mod.HOST_LIB) and os.path.exists(mod.HIP_LIB)):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def golden():
    return {n: json.load(open(os.path.join(GOLDEN, n + ".json"))) for n in ("flatten", "camera", "frames", "terrain", "color", "room")}


@pytest.fixture(scope="session")
def product_scenes(V):
    """map name -> (texels, tex_dim) flattened by the PRODUCT host library."""
    out = {}
    for m in ("dragon", "monu9", "nature"):
        w = V.World()
        assert w.load_vox(os.path.join(MAPS, m + ".vox"))
        out[m] = w.flatten()
        w.close()
    out["terrain"] = terrain_world(V).flatten()
    out["room"] = room_world(V).flatten()
    return out


def room_world(V):
    """The reference's translucent room (src/main.cpp:505-633 as the ordered insert list tests/golden/room.npz, written by
    make_room.py) built by the PRODUCT host library"""
    d = np.load(os.path.join(GOLDEN, "room.npz"))
    mats = json.load(open(os.path.join(GOLDEN, "room.json")))["materials"]
    w = V.World()
    for (x, y, z), c, m in zip(d["xyz"], d["color"], d["material"]):
        mm = mats[int(m)]
        w.insert(int(x), int(y), int(z), int(c), mm["refraction"], mm["illumination"], mm["k"])
    return w


def room_tree(O):
    """the same insert list into an ORACLE tree"""
    d = np.load(os.path.join(GOLDEN, "room.npz"))
    mats = json.load(open(os.path.join(GOLDEN, "room.json")))["materials"]
    t = O.new_tree()
    L = O.lib()
    for (x, y, z), c, m in zip(d["xyz"], d["color"], d["material"]):
        mm = mats[int(m)]
        L.o_octree_insert(t, O.VoxelObj(O.IVec3(int(x), int(y), int(z)), int(c), O.Voxel(mm["refraction"], mm["illumination"], mm["k"])))
    return t


def terrain_world(V, window=None):
    """BASELINE config 4 built by the PRODUCT host library: tests/golden/terrain.json over terrain_heights.npz"""
    t = json.load(open(os.path.join(GOLDEN, "terrain.json")))
    wd = window or t["window"]
    w = V.World()
    w.fill_heights(np.load(os.path.join(GOLDEN, "terrain_heights.npz"))["heights"], wd["x0"], wd["z0"], wd["nx"], wd["nz"],
                   t["band"], t["floor"])
    return w


def random_voxels(rng, n, lo, hi, n_colors=5):
    xyz = rng.integers(lo, hi, size=(n, 3), dtype=np.int32)
    palette = np.array([0x50b43cff, 0x644628ff, 0xa0a0a0ff, 0x3c64dc96, 0xffd2d2ff, 0x112233ff, 0xc8dcff50], np.uint32)
    rgba = palette[rng.integers(0, n_colors, size=n)]
    return xyz, rgba
