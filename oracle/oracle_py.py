"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_FULL = 0, 1, 2


class IVec3(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("z", C.c_int32)]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Voxel(C.Structure):
    _fields_ = [("refraction", C.c_float), ("illumination", C.c_float), ("k", C.c_float)]


class VoxelObj(C.Structure):
    _fields_ = [("coord", IVec3), ("color", C.c_uint32), ("voxel", Voxel)]


class Octree(C.Structure):
    pass


Octree._fields_ = [("voxel", VoxelObj), ("has_voxel", C.c_int), ("children", C.POINTER(C.POINTER(Octree))),
                   ("parent", C.POINTER(Octree)), ("lbb", IVec3), ("rtf", IVec3)]


class Camera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("front", C.c_float * 3), ("up", C.c_float * 3),
                ("right", C.c_float * 3), ("world_up", C.c_float * 3), ("yaw", C.c_float), ("pitch", C.c_float)]


class Scene(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("n_texels", C.c_size_t), ("tex_dim", C.c_int32),
                ("voxel_scale", C.c_float), ("bounds_min", C.c_int32 * 3), ("bounds_max", C.c_int32 * 3),
                ("global_light", C.c_float * 4), ("light_dir", C.c_float * 3), ("highlighted", C.c_int32 * 3),
                ("inv_proj", C.c_float * 16), ("inv_view", C.c_float * 16), ("cam_pos", C.c_float * 4),
                ("wide_pointers", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("fetches", C.c_uint64), ("finds", C.c_uint64), ("root_restarts", C.c_uint64),
                ("steps", C.c_uint64), ("hits", C.c_uint64), ("shadow_rays", C.c_uint64)]


def build(force=False):
    if force or not os.path.exists(LIB) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB)
            for f in os.listdir(HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.o_octree_create.restype = C.POINTER(Octree)
        L.o_octree_create.argtypes = [C.POINTER(Octree), IVec3, IVec3]
        L.o_octree_insert.argtypes = [C.POINTER(Octree), VoxelObj]
        L.o_octree_remove.argtypes = [C.POINTER(Octree), IVec3]
        L.o_octree_find.restype = VoxelObj
        L.o_octree_find.argtypes = [C.POINTER(Octree), IVec3]
        L.o_octree_delete.argtypes = [C.POINTER(Octree)]
        L.o_octree_texel_size.restype = C.c_size_t
        L.o_octree_texel_size.argtypes = [C.POINTER(Octree)]
        L.o_octree_texture.restype = C.POINTER(C.c_uint8)
        L.o_octree_texture.argtypes = [C.POINTER(Octree), C.POINTER(C.c_size_t), C.c_size_t]
        L.o_octree_texture_wide.restype = C.POINTER(C.c_uint8)
        L.o_octree_texture_wide.argtypes = [C.POINTER(Octree), C.POINTER(C.c_size_t)]
        L.o_octree_ray_cast.restype = C.POINTER(Octree)
        L.o_octree_ray_cast.argtypes = [C.POINTER(Octree), Vec3, Vec3, Vec3, Vec3]
        L.o_fill_heights.argtypes = [C.POINTER(Octree), C.c_void_p] + [C.c_int] * 8
        L.o_tex_dim_for.restype = C.c_uint32
        L.o_tex_dim_for.argtypes = [C.c_size_t]
        L.o_load_vox_mem.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Octree), C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_long)]
        L.o_load_vox_file.argtypes = [C.c_char_p, C.POINTER(Octree), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
        L.o_camera_init.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.c_float, C.c_float]
        L.o_camera_ubo.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float)]
        L.o_scene_defaults.argtypes = [C.POINTER(Scene)]
        L.o_render.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.POINTER(Stats)]
        L.o_find_point.argtypes = [C.POINTER(Scene), C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.o_fnv1a64.restype = C.c_uint64
        L.o_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        for n in ("o_det_expf", "o_det_sinf", "o_det_cosf"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.o_det_powf.restype = C.c_float
        L.o_det_powf.argtypes = [C.c_float, C.c_float]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


WORLD_MIN, WORLD_MAX = (-1023, -1023, -1023), (1024, 1024, 1024)  # src/main.cpp:478-480


def new_tree():
    return lib().o_octree_create(None, IVec3(*WORLD_MIN), IVec3(*WORLD_MAX))


def flatten(tree, wide=False):
    """-> (uint8 ndarray of texel bytes (may be empty), tex_dim). wide: the NON-REFERENCE extension stream of
    o_octree_texture_wide() (31-bit child addresses) for trees beyond 2^23 texels; render it with make_scene(wide=True)."""
    L = lib()
    n = L.o_octree_texel_size(tree)
    dim = L.o_tex_dim_for(n)
    sz = C.c_size_t(0)
    p = L.o_octree_texture_wide(tree, C.byref(sz)) if wide else L.o_octree_texture(tree, C.byref(sz), dim)
    if not p:
        return np.zeros(0, np.uint8), int(dim)
    arr = np.ctypeslib.as_array(p, shape=(sz.value,)).copy()
    L.free(p)
    return arr, int(dim)


def fill_heights(tree, heights, x0, z0, nx, nz, band=8, floor_y=20):
    """config 4 terrain (src/main.cpp:487-503 over a uint16 height field [size_z, size_x]) into `tree`"""
    h = np.ascontiguousarray(heights, dtype=np.uint16)
    if lib().o_fill_heights(tree, h.ctypes.data, h.shape[1], h.shape[0], x0, z0, nx, nz, band, floor_y) != 0:
        raise ValueError("o_fill_heights: bad arguments")


def load_vox(path_or_bytes, offset=(0, 0, 0)):
    """-> (tree, ok, n_inserted)"""
    L = lib()
    t = new_tree()
    n = C.c_long(0)
    if isinstance(path_or_bytes, (bytes, bytearray)):
        ok = L.o_load_vox_mem(bytes(path_or_bytes), len(path_or_bytes), t, *offset, C.byref(n))
    else:
        ok = L.o_load_vox_file(str(path_or_bytes).encode(), t, *offset, C.byref(n))
    return t, bool(ok), n.value


def camera_ubo(pos, yaw, pitch, width, height):
    L = lib()
    cam = Camera()
    L.o_camera_init(C.byref(cam), (C.c_float * 3)(*pos), C.c_float(yaw), C.c_float(pitch))
    ip, iv, cp = (C.c_float * 16)(), (C.c_float * 16)(), (C.c_float * 4)()
    L.o_camera_ubo(C.byref(cam), width, height, ip, iv, cp)
    return (np.array(ip, np.float32), np.array(iv, np.float32), np.array(cp, np.float32)), cam


def make_scene(texels, tex_dim, inv_proj, inv_view, cam_pos, highlighted=(-1, -1, -1), wide=False):
    L = lib()
    s = Scene()
    L.o_scene_defaults(C.byref(s))
    texels = np.ascontiguousarray(texels, np.uint8)
    s._keep = texels
    s.texels = texels.ctypes.data if texels.size else None
    s.n_texels = texels.size // 4
    s.tex_dim = int(tex_dim)
    s.inv_proj[:] = [float(x) for x in inv_proj]
    s.inv_view[:] = [float(x) for x in inv_view]
    s.cam_pos[:] = [float(x) for x in cam_pos]
    s.highlighted[:] = list(highlighted)
    s.wide_pointers = 1 if wide else 0
    return s


def render(scene, width, height, mode, row0=0, row1=None, want_fetch_map=False):
    L = lib()
    row1 = height if row1 is None else row1
    rgba = np.zeros((height, width, 4), np.uint8)
    idd = np.zeros((height, width, 2), np.int32)
    fm = np.zeros((height, width), np.uint32) if want_fetch_map else None
    st = Stats()
    L.o_render(C.byref(scene), width, height, row0, row1, mode, rgba.ctypes.data, idd.ctypes.data,
               fm.ctypes.data if fm is not None else None, C.byref(st))
    stats = {k: getattr(st, k) for k, _ in Stats._fields_}
    return rgba, idd, fm, stats


def denoise(rgba, id_dist):
    """shaders/quad.frag restatement -> rgba8[H,W,4]"""
    L = lib()
    L.o_denoise.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rgba = np.ascontiguousarray(rgba, np.uint8)
    idd = np.ascontiguousarray(id_dist, np.int32)
    h, w = rgba.shape[:2]
    out = np.zeros_like(rgba)
    L.o_denoise(rgba.ctypes.data, idd.ctypes.data, w, h, out.ctypes.data)
    return out


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().o_fnv1a64(a.ctypes.data, a.nbytes))
