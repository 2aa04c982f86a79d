"""ctypes bindings over the two C-ABI libraries of the MI355X ray-casting path.

    libvrt_host.so  (include/vrt_host.h)  host API kept from the reference:
                    octree build/flatten, .vox loader, camera block
    libvrt_hip.so   (include/vrt.h)       the gfx950 dispatch layer

The directory name carries a hyphen, so import it with
``importlib.import_module("voxel-raytracer_amd")`` (see ``vrt_import.py`` at the
repository root). There is no CPU fallback anywhere in this package: `Context`
raises if the HIP library is missing or no GPU is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.environ.get("VRT_HIP_LIB") or os.path.join(HERE, "libvrt_hip.so")   # VRT_HIP_LIB: an A/B build of the same library (tools)
HOST_LIB = os.path.join(HERE, "libvrt_host.so")
TEST_LIB = os.path.join(HERE, "libvrt_hip_test.so")   # test support, not product (csrc/test/vrt_test.hip)
OPT_RAY_TABLES, OPT_EMPTY_OCTANTS, OPT_DISPLAY_KERNEL, OPT_FULL_OPAQUE, OPT_HEAVY_TILES = 1, 2, 3, 4, 5

MODE_PRIMARY, MODE_PRIMARY_SHADOW, MODE_FULL = 0, 1, 2
MODES = {"primary": MODE_PRIMARY, "primary_shadow": MODE_PRIMARY_SHADOW, "full": MODE_FULL}


class VrtError(RuntimeError):
    pass


def build(targets=("../libvrt_hip.so", "../libvrt_hip_test.so", "../libvrt_host.so")):
    """Compile the libraries in-tree (hipcc --offload-arch=gfx950 / g++)."""
    subprocess.check_call(["make", "-j8", "-C", CSRC, *targets])


class Params(C.Structure):
    _fields_ = [("voxel_scale", C.c_float), ("world_min", C.c_int32 * 3), ("world_max", C.c_int32 * 3),
                ("global_light", C.c_float * 4), ("light_dir", C.c_float * 3), ("highlighted", C.c_int32 * 3)]


class View(C.Structure):
    """vrt_view: one camera block and the two device images it renders into"""
    _fields_ = [("inv_projection", C.c_float * 16), ("inv_view", C.c_float * 16), ("camera_pos", C.c_float * 4),
                ("d_rgba8", C.c_void_p), ("d_id_dist", C.c_void_p)]


class Patch(C.Structure):
    """vrt_patch: an ancestor of an edited voxel, by depth and child-index path"""
    _fields_ = [("depth", C.c_int32), ("path", C.c_uint8 * 16)]


class SceneInfo(C.Structure):
    _fields_ = [("tex_dim", C.c_uint32), ("n_texels", C.c_uint32), ("n_records", C.c_uint32),
                ("n_internal", C.c_uint32), ("n_leaves", C.c_uint32), ("max_depth", C.c_uint32),
                ("lds_records", C.c_uint32), ("reserved", C.c_uint32)]


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise VrtError(f"{HOST_LIB} is missing: run __graft_entry__.build() (make -C {CSRC})")
        L = C.CDLL(HOST_LIB)
        L.vrth_world_create.restype = C.c_void_p
        L.vrth_world_create.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.vrth_world_destroy.argtypes = [C.c_void_p]
        L.vrth_world_load_vox.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.vrth_world_load_vox_mem.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_long)]
        L.vrth_world_insert.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_float, C.c_float,
                                        C.c_float]
        L.vrth_world_insert_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                             C.c_float]
        L.vrth_world_remove.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.vrth_world_find.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
        L.vrth_world_ray_cast.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int)]
        L.vrth_world_ray_cast_many.restype = C.c_long
        L.vrth_world_ray_cast_many.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.vrth_world_texel_count.restype = C.c_size_t
        L.vrth_world_texel_count.argtypes = [C.c_void_p]
        L.vrth_world_flatten.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_uint32)]
        L.vrth_free.argtypes = [C.c_void_p]
        L.vrth_camera_block.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_int, C.c_int,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float)]
        L.vrth_write_vox.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vrth_encode_vox.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.vrth_make_custom_vox.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.vrth_world_fill_heights.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 8
        L.vrth_fnv1a64.restype = C.c_uint64
        L.vrth_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        L.vrth_version.restype = C.c_char_p
        _host = L
    return _host


def hip_lib():
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise VrtError(f"{HIP_LIB} is missing: the HIP extension must be built (make -C {CSRC}); "
                           "there is no CPU fallback")
        # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7). Loading it
        # first lets libvrt_hip.so bind to that copy; the other order ends with two HIP/HSA runtimes and
        # torch reporting "No HIP GPUs are available".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(HIP_LIB)
        L.vrt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.vrt_destroy.argtypes = [C.c_void_p]
        L.vrt_last_error.restype = C.c_char_p
        L.vrt_last_error.argtypes = [C.c_void_p]
        L.vrt_default_params.argtypes = [C.POINTER(Params)]
        L.vrt_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.vrt_upload_octree.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.vrt_get_scene_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.vrt_set_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vrt_dispatch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vrt_dispatch_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.vrt_dispatch_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_shard_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.vrt_dispatch_timed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.vrt_synchronize.argtypes = [C.c_void_p]
        L.vrt_denoise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_denoise_host.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vrt_get_tile_order.restype = C.c_long
        L.vrt_get_tile_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.vrt_set_tile_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.vrt_dispatch_views.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(View), C.c_int, C.c_void_p]
        L.vrt_patch_plan.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Patch)]
        L.vrt_patch_apply.argtypes = [C.c_void_p, C.POINTER(Patch), C.c_void_p, C.c_size_t]
        L.vrt_dispatch_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.vrt_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.vrt_stream.restype = C.c_void_p
        L.vrt_stream.argtypes = [C.c_void_p]
        L.vrt_device.argtypes = [C.c_void_p]
        L.vrt_set_variant.argtypes = [C.c_void_p, C.c_int]
        L.vrt_variant_available.argtypes = [C.c_int]
        L.vrt_patch_begin.argtypes = [C.c_void_p]
        L.vrt_patch_end.argtypes = [C.c_void_p]
        L.vrt_compact.argtypes = [C.c_void_p]
        L.vrt_dispatch_async.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.vrt_dispatch_wait.argtypes = [C.c_void_p, C.c_int]
        L.vrt_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.vrt_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.vrt_dispatch_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_device_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.vrt_device_free.argtypes = [C.c_void_p, C.c_void_p]
        L.vrt_device_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vrt_device_write.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vrt_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.vrt_ipc_export.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.vrt_ipc_open.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.vrt_ipc_close.argtypes = [C.c_void_p, C.c_void_p]
        L.vrt_stream_write_flag.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.vrt_stream_wait_flag.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.vrt_create_multi.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.vrt_destroy_multi.argtypes = [C.c_void_p]
        L.vrt_multi_last_error.restype = C.c_char_p
        L.vrt_multi_last_error.argtypes = [C.c_void_p]
        L.vrt_multi_devices.argtypes = [C.c_void_p]
        L.vrt_multi_context.restype = C.c_void_p
        L.vrt_multi_context.argtypes = [C.c_void_p, C.c_int]
        L.vrt_multi_upload_octree.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.vrt_multi_set_camera.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vrt_multi_set_params.argtypes = [C.c_void_p, C.POINTER(Params)]
        L.vrt_multi_frame_alloc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.vrt_multi_frame_free.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_multi_dispatch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vrt_multi_synchronize.argtypes = [C.c_void_p]
        L.vrt_multi_stream.restype = C.c_void_p
        L.vrt_multi_stream.argtypes = [C.c_void_p]
        L.vrt_version.restype = C.c_char_p
        if hasattr(L, "vrt_ab_set_full_split"):   # `make AB=1` builds only
            L.vrt_ab_set_full_split.argtypes = [C.c_void_p, C.c_int]
            L.vrt_ab_set_bounce.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _hip = L
    return _hip


_test = None


def test_lib():
    """libvrt_hip_test.so: device probes of the kernels' arithmetic and host-only views of the uploader's layouts, for the
    parity suite (csrc/test/vrt_test.hip). Not part of the product; libvrt_hip.so exports none of it."""
    global _test
    if _test is None:
        if not os.path.exists(TEST_LIB):
            raise VrtError(f"{TEST_LIB} is missing: make -C {CSRC}")
        try:
            import torch  # noqa: F401  (one HIP runtime per process, see hip_lib())
        except ImportError:
            pass
        L = C.CDLL(TEST_LIB)
        L.vrt_test_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vrt_test_build_layout.restype = C.c_long
        L.vrt_test_build_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(SceneInfo)]
        L.vrt_test_patch_check.restype = C.c_long
        L.vrt_test_patch_check.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.vrt_test_ray_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vrt_test_root0.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]
        L.vrt_test_view_in_range.argtypes = [C.c_void_p]
        L.vrt_test_wide_find.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p,
                                         C.c_size_t, C.c_void_p, C.c_void_p]
        _test = L
    return _test


def available_variants(upto=64):
    """kernel variants compiled into libvrt_hip.so (the shipped ones; all of them in a `make AB=1` build)"""
    L = hip_lib()
    return [v for v in range(upto) if L.vrt_variant_available(v)]


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class World:
    """An octree root plus the host API calls around it (octree.hpp / voxReader.hpp)."""

    def __init__(self, world_min=None, world_max=None):
        L = host_lib()
        mn = (C.c_int32 * 3)(*world_min) if world_min is not None else None
        mx = (C.c_int32 * 3)(*world_max) if world_max is not None else None
        self._h = L.vrth_world_create(mn, mx)
        if not self._h:
            raise VrtError("vrth_world_create failed")

    def close(self):
        if getattr(self, "_h", None):
            try:
                host_lib().vrth_world_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None

    __del__ = close

    def load_vox(self, path, offset=(0, 0, 0)):
        return bool(host_lib().vrth_world_load_vox(self._h, str(path).encode(), *offset))

    def load_vox_bytes(self, data, offset=(0, 0, 0)):
        n = C.c_long(0)
        ok = host_lib().vrth_world_load_vox_mem(self._h, bytes(data), len(data), *offset, C.byref(n))
        return bool(ok), n.value

    def insert(self, x, y, z, rgba, refraction=3.0, illumination=0.0, k=0.0):
        host_lib().vrth_world_insert(self._h, x, y, z, rgba, refraction, illumination, k)

    def insert_many(self, xyz, rgba, refraction=3.0, illumination=0.0, k=0.0):
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        rgba = np.ascontiguousarray(rgba, np.uint32).reshape(-1)
        assert xyz.shape[0] == rgba.shape[0]
        host_lib().vrth_world_insert_many(self._h, xyz.ctypes.data, rgba.ctypes.data, xyz.shape[0], refraction,
                                          illumination, k)

    def remove(self, x, y, z):
        host_lib().vrth_world_remove(self._h, x, y, z)

    def find(self, x, y, z):
        out = (C.c_uint32 * 7)()
        host_lib().vrth_world_find(self._h, x, y, z, out)
        f = np.array(out[4:7], np.uint32).view(np.float32)
        return {"coord": tuple(int(np.int32(np.uint32(v))) for v in out[0:3]), "color": int(out[3]),
                "refraction": float(f[0]), "illumination": float(f[1]), "k": float(f[2])}

    def ray_cast(self, origin, direction):
        o = (C.c_float * 3)(*origin)
        d = (C.c_float * 3)(*direction)
        hit = (C.c_int32 * 3)()
        has = C.c_int(0)
        r = host_lib().vrth_world_ray_cast(self._h, o, d, hit, C.byref(has))
        return (tuple(hit), bool(has.value)) if r == 1 else None

    def ray_cast_many(self, origin, dirs):
        """octree_ray_cast for every direction of dirs[n, 3] from one origin -> (hit uint8[n]: 0 miss / 1 voxel / 2 node, coords int32[n, 3])"""
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        hit = np.zeros(d.shape[0], np.uint8)
        coords = np.zeros((d.shape[0], 3), np.int32)
        if host_lib().vrth_world_ray_cast_many(self._h, (C.c_float * 3)(*origin), d.ctypes.data, d.shape[0], hit.ctypes.data,
                                               coords.ctypes.data) < 0:
            raise VrtError("vrth_world_ray_cast_many failed")
        return hit, coords

    def texel_count(self):
        return host_lib().vrth_world_texel_count(self._h)

    def flatten(self):
        """-> (uint8 ndarray with the octree_texture() bytes, tex_dim)"""
        p = C.c_void_p()
        n = C.c_size_t(0)
        d = C.c_uint32(0)
        if host_lib().vrth_world_flatten(self._h, C.byref(p), C.byref(n), C.byref(d)) != 0:
            raise VrtError("vrth_world_flatten failed")
        if not p.value:
            return np.zeros(0, np.uint8), int(d.value)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()
        host_lib().vrth_free(p)
        return arr, int(d.value)

    def records(self):
        """EXTENSION: the device record array straight from the tree -> (uint32[n,2], tex_dim), or None when the
        root is itself a leaf (texel path only)."""
        L = host_lib()
        L.vrth_world_records.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
        p, n, d = C.c_void_p(), C.c_size_t(0), C.c_uint32(0)
        r = L.vrth_world_records(self._h, C.byref(p), C.byref(n), C.byref(d))
        if r == -2:
            return None
        if r != 0:
            raise VrtError("vrth_world_records failed")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 2)).copy()
        L.vrth_free(p)
        return arr, int(d.value)

    def box_records(self, path, lo, hi):
        """EXTENSION: the sub-tree under the node reached by `path` (child indices from the root) as device records, only the nodes
        that meet the box of voxels [lo, hi] (inclusive) expanded, the rest "keep" records (vrth_world_box_records) -> uint32[n, 2]"""
        L = host_lib()
        L.vrth_world_box_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        p, n = C.c_void_p(), C.c_size_t(0)
        pa = (C.c_uint8 * 16)(*list(path))
        r = L.vrth_world_box_records(self._h, pa, len(path), (C.c_int32 * 3)(*[int(v) for v in lo]), (C.c_int32 * 3)(*[int(v) for v in hi]),
                                     C.byref(p), C.byref(n))
        if r != 0:
            raise VrtError(f"vrth_world_box_records failed ({r})")
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value, 2)).copy()
        L.vrth_free(p)
        return arr

    def fill_heights(self, heights, x0, z0, nx, nz, band=8, floor_y=20):
        """config 4 terrain: the reference's generator (src/main.cpp:487-503) over a uint16 height field [size_z, size_x]"""
        h = np.ascontiguousarray(heights, dtype=np.uint16)
        if host_lib().vrth_world_fill_heights(self._h, h.ctypes.data, h.shape[1], h.shape[0], x0, z0, nx, nz, band, floor_y) != 0:
            raise VrtError("vrth_world_fill_heights failed")


def camera_block(pos, yaw, pitch, width, height):
    """Camera(pos, up, yaw, pitch) -> (inv_projection[16], inv_view[16], camera_pos[4], front[3]) float32."""
    ip, iv = np.zeros(16, np.float32), np.zeros(16, np.float32)
    cp, fr = np.zeros(4, np.float32), np.zeros(3, np.float32)
    p = (C.c_float * 3)(*pos)
    if host_lib().vrth_camera_block(p, yaw, pitch, width, height, _fptr(ip), _fptr(iv), _fptr(cp), _fptr(fr)) != 0:
        raise VrtError("vrth_camera_block failed")
    return ip, iv, cp, fr


def encode_vox(size, xyzi, palette=None):
    xyzi = np.ascontiguousarray(xyzi, np.uint8).reshape(-1, 4)
    pal = np.ascontiguousarray(palette, np.uint8).reshape(256, 4) if palette is not None else None
    out, n = C.c_void_p(), C.c_size_t(0)
    r = host_lib().vrth_encode_vox(size[0], size[1], size[2], xyzi.ctypes.data, xyzi.shape[0],
                                   pal.ctypes.data if pal is not None else None, C.byref(out), C.byref(n))
    if r != 0:
        raise VrtError("vrth_encode_vox failed")
    data = C.string_at(out, n.value)
    host_lib().vrth_free(out)
    return data


def make_custom_vox():
    out, n = C.c_void_p(), C.c_size_t(0)
    if host_lib().vrth_make_custom_vox(C.byref(out), C.byref(n)) != 0:
        raise VrtError("vrth_make_custom_vox failed")
    data = C.string_at(out, n.value)
    host_lib().vrth_free(out)
    return data


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(host_lib().vrth_fnv1a64(a.ctypes.data, a.nbytes))


def build_layout(texels):
    """Host-only: the device record array the uploader would build -> (uint32[n,2], SceneInfo)."""
    L = test_lib()
    t = np.ascontiguousarray(texels, np.uint8)
    info = SceneInfo()
    n = L.vrt_test_build_layout(t.ctypes.data if t.size else None, t.size, None, 0, C.byref(info))
    if n < 0:
        raise VrtError("malformed texel stream")
    rec = np.zeros((n, 2), np.uint32)
    L.vrt_test_build_layout(t.ctypes.data if t.size else None, t.size, rec.ctypes.data, n, None)
    return rec, info


def patch_check(texels_before, texels_after, voxel, points, world_min=(-1023, -1023, -1023), world_max=(1024, 1024, 1024),
                sparse=False):
    """Host-only (vrt_test_patch_check): patch the layouts of the tree before an edit of `voxel` with the sub-tree of
    the tree after it and compare point lookups with freshly built layouts.
    -> (mismatching points, depth of the node replaced or 0, records appended, wide cells appended, texel count ok)"""
    L = test_lib()
    tb = np.ascontiguousarray(texels_before, np.uint8)
    ta = np.ascontiguousarray(texels_after, np.uint8)
    pts = np.ascontiguousarray(points, np.int32).reshape(-1, 3)
    info = np.zeros(4, np.uint32)
    r = L.vrt_test_patch_check(tb.ctypes.data if tb.size else None, tb.size, ta.ctypes.data if ta.size else None, ta.size,
                                (C.c_int32 * 3)(*world_min), (C.c_int32 * 3)(*world_max), int(voxel[0]), int(voxel[1]),
                                int(voxel[2]), pts.ctypes.data, pts.shape[0], info.ctypes.data, 1 if sparse else 0)
    if r < 0:
        raise VrtError(f"vrt_test_patch_check failed ({r})")
    return int(r), int(info[0]), int(info[1]), int(info[2]), bool(info[3])


def ray_table(inv_projection, width, height):
    """Host-only: the per-column / per-row ray-generation table of a projection (vrt_test_ray_table)
    -> (x[width], y[height], z) float32, or None when the projection has no table."""
    L = test_lib()
    m = np.ascontiguousarray(inv_projection, np.float32).reshape(16)
    x, y, z = np.zeros(width, np.float32), np.zeros(height, np.float32), np.zeros(1, np.float32)
    r = L.vrt_test_ray_table(m.ctypes.data, width, height, x.ctypes.data, y.ctypes.data, z.ctypes.data)
    if r < 0:
        raise VrtError(f"vrt_test_ray_table failed ({r})")
    return (x, y, float(z[0])) if r == 1 else None


def root0_choice(texels, eye, world_min=(-1023, -1023, -1023), world_max=(1024, 1024, 1024)):
    """Host-only (vrt_test_root0): (root0_only, log2 side of the root chosen for this eye, its minimum corner (3), log2 side
    of build_wide()'s root), or None when the scene has no wide form."""
    L = test_lib()
    t = np.ascontiguousarray(texels, np.uint8)
    out = (C.c_int32 * 6)()
    r = L.vrt_test_root0(t.ctypes.data if t.size else None, t.size, (C.c_int32 * 3)(*world_min), (C.c_int32 * 3)(*world_max),
                          (C.c_int32 * 3)(*[int(v) for v in eye]), out)
    if r == -5:
        return None
    if r != 0:
        raise VrtError(f"vrt_test_root0 failed ({r})")
    return bool(out[0]), int(out[1]), (int(out[2]), int(out[3]), int(out[4])), int(out[5])

def test_tile_order(tile_ticks, wave_slots, device=0):
    """Device probe (vrt_test_tile_order): the feedback scheduler's order kernel on synthetic per-tile ticks (4 per group) ->
    (order of the groups, how many groups at its head the general full path tracer would trace as part-tile waves)."""
    L = test_lib()
    t = np.ascontiguousarray(tile_ticks, np.uint32).reshape(-1)
    assert t.size % 4 == 0
    n = t.size // 4
    L.vrt_test_tile_order.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    order = np.zeros(n, np.uint32)
    split = np.zeros(1, np.uint32)
    r = L.vrt_test_tile_order(device, t.ctypes.data, n, wave_slots, order.ctypes.data, split.ctypes.data)
    if r:
        raise VrtError(f"vrt_test_tile_order failed ({r})")
    return order, int(split[0])



def tree_is_opaque(texels):
    """Host-only (vrt_test_tree_is_opaque): may VRT_MODE_FULL run without a ray stack on this tree (for an eye in empty space)?"""
    L = test_lib()
    L.vrt_test_tree_is_opaque.argtypes = [C.c_void_p, C.c_size_t]
    t = np.ascontiguousarray(texels, np.uint8)
    r = L.vrt_test_tree_is_opaque(t.ctypes.data if t.size else None, t.size)
    if r < 0:
        raise VrtError(f"vrt_test_tree_is_opaque failed ({r})")
    return r == 1


def view_in_range(inv_view):
    m = np.ascontiguousarray(inv_view, np.float32).reshape(16)
    return test_lib().vrt_test_view_in_range(m.ctypes.data) == 1


def wide_find(texels, points, world_min=(-1023, -1023, -1023), world_max=(1024, 1024, 1024)):
    """Host-only: point queries through the wide (64-cell) layout the default kernels read.
    -> (uint32[n,8] = w0, w1, mn[3], mx[3], (wide nodes, roots)), or None when the scene has no wide form."""
    L = test_lib()
    t = np.ascontiguousarray(texels, np.uint8)
    pts = np.ascontiguousarray(points, np.int32).reshape(-1, 3)
    out = np.zeros((pts.shape[0], 8), np.uint32)
    stats = np.zeros(2, np.uint32)
    r = L.vrt_test_wide_find(t.ctypes.data if t.size else None, t.size, (C.c_int32 * 3)(*world_min),
                              (C.c_int32 * 3)(*world_max), pts.ctypes.data, pts.shape[0], out.ctypes.data,
                              stats.ctypes.data)
    if r == -5:
        return None
    if r != 0:
        raise VrtError(f"vrt_test_wide_find failed ({r})")
    return out, (int(stats[0]), int(stats[1]))


class Context:
    """One GPU's dispatch context (vrt_ctx)."""

    def __init__(self, device=0):
        L = hip_lib()
        h = C.c_void_p()
        r = L.vrt_create(device, C.byref(h))
        if r != 0:
            raise VrtError(f"vrt_create({device}) failed ({r}): {L.vrt_last_error(None).decode()}")
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "borrowed", False):
            self._L.vrt_destroy(self._h)
        self._h = None

    __del__ = close

    def _chk(self, r):
        if r != 0:
            raise VrtError(f"vrt error {r}: {self._L.vrt_last_error(self._h).decode()}")

    def upload_octree(self, texels, tex_dim):
        t = np.ascontiguousarray(texels, np.uint8)
        self._chk(self._L.vrt_upload_octree(self._h, t.ctypes.data if t.size else None, t.size, tex_dim))

    def upload_records(self, records, tex_dim):
        """EXTENSION: upload the record array World.records() returns (no texel stream, no 2^23-texel limit)."""
        r = np.ascontiguousarray(records, np.uint32).reshape(-1, 2)
        self._L.vrt_upload_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        self._chk(self._L.vrt_upload_records(self._h, r.ctypes.data, r.shape[0], tex_dim))

    def scene_info(self):
        info = SceneInfo()
        self._chk(self._L.vrt_get_scene_info(self._h, C.byref(info)))
        return {k: getattr(info, k) for k, _ in SceneInfo._fields_}

    def set_camera(self, inv_proj, inv_view, cam_pos):
        ip = np.ascontiguousarray(inv_proj, np.float32)
        iv = np.ascontiguousarray(inv_view, np.float32)
        cp = np.ascontiguousarray(cam_pos, np.float32)
        self._chk(self._L.vrt_set_camera(self._h, _fptr(ip), _fptr(iv), _fptr(cp)))

    def default_params(self):
        p = Params()
        self._L.vrt_default_params(C.byref(p))
        return p

    def set_params(self, p):
        self._chk(self._L.vrt_set_params(self._h, C.byref(p)))

    def set_variant(self, v):
        self._chk(self._L.vrt_set_variant(self._h, v))

    def dispatch(self, width, height, mode=MODE_PRIMARY):
        """Synchronous frame into host arrays -> (rgba8[H,W,4] uint8, id_dist[H,W,2] int32)."""
        rgba = np.zeros((height, width, 4), np.uint8)
        idd = np.zeros((height, width, 2), np.int32)
        self._chk(self._L.vrt_dispatch(self._h, width, height, mode, rgba.ctypes.data, idd.ctypes.data))
        return rgba, idd

    def dispatch_rows(self, width, height, row_begin, row_end, mode, d_rgba, d_id, stream=None):
        self._chk(self._L.vrt_dispatch_rows(self._h, width, height, row_begin, row_end, mode, d_rgba, d_id, stream))

    def dispatch_shard(self, width, height, tile_rows, shard, n_shards, mode, d_rgba, d_id, stream=None):
        self._chk(self._L.vrt_dispatch_shard(self._h, width, height, tile_rows, shard, n_shards, mode, d_rgba, d_id,
                                             stream))

    def host_alloc(self, shape, dtype):
        """a page-locked numpy array (vrt_host_alloc); release with host_free(arr)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._chk(self._L.vrt_host_alloc(self._h, n, C.byref(p)))
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).view(dtype).reshape(shape)
        return arr

    def host_free(self, arr):
        self._chk(self._L.vrt_host_free(self._h, arr.ctypes.data))

    def dispatch_async(self, width, height, mode, out_rgba, out_id):
        t = C.c_int(-1)
        self._chk(self._L.vrt_dispatch_async(self._h, width, height, mode, out_rgba.ctypes.data if out_rgba is not None else None,
                                             out_id.ctypes.data if out_id is not None else None, C.byref(t)))
        return t.value

    def dispatch_wait(self, ticket):
        self._chk(self._L.vrt_dispatch_wait(self._h, ticket))

    def dispatch_tiles(self, width, height, tile_rows, shard, n_shards, mode, d_frame_rgba, d_frame_id, stream=None):
        """the shard's row tiles at their place in a FULL frame (local, peer or IPC-mapped device memory)"""
        self._chk(self._L.vrt_dispatch_tiles(self._h, width, height, tile_rows, shard, n_shards, mode, d_frame_rgba, d_frame_id, stream))

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self._L.vrt_device_alloc(self._h, nbytes, C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self._chk(self._L.vrt_device_free(self._h, ptr))

    def device_read(self, ptr, shape, dtype, stream=None):
        out = np.empty(shape, dtype)
        self._chk(self._L.vrt_device_read(self._h, ptr, out.ctypes.data, out.nbytes, stream))
        return out

    def device_write(self, ptr, arr, stream=None):
        a = np.ascontiguousarray(arr)
        self._chk(self._L.vrt_device_write(self._h, ptr, a.ctypes.data, a.nbytes, stream))

    def device_copy(self, dst, src, nbytes, stream=None):
        """asynchronous device-to-device copy on `stream`; either side may be an IPC mapping of another rank's memory"""
        self._chk(self._L.vrt_device_copy(self._h, C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), C.c_void_p(stream)))

    def ipc_export(self, ptr):
        h = C.create_string_buffer(64)
        self._chk(self._L.vrt_ipc_export(self._h, ptr, h))
        return h.raw

    def ipc_open(self, handle):
        p = C.c_void_p()
        self._chk(self._L.vrt_ipc_open(self._h, bytes(handle), C.byref(p)))
        return p.value

    def ipc_close(self, ptr):
        self._chk(self._L.vrt_ipc_close(self._h, ptr))

    def stream_write_flag(self, d_flag, value, stream=None):
        self._chk(self._L.vrt_stream_write_flag(self._h, d_flag, value, stream))

    def stream_wait_flag(self, d_flag, value, stream=None):
        self._chk(self._L.vrt_stream_wait_flag(self._h, d_flag, value, stream))

    def dispatch_timed(self, width, height, row_begin, row_end, mode, d_rgba, d_id, iters, stream=None):
        ms = (C.c_float * iters)()
        self._chk(self._L.vrt_dispatch_timed(self._h, width, height, row_begin, row_end, mode, d_rgba, d_id, stream,
                                             iters, ms))
        return np.array(ms, np.float32)

    def set_profiling(self, max_launches, every=1):
        self._L.vrt_set_profiling_stride.argtypes = [C.c_void_p, C.c_int]
        self._chk(self._L.vrt_set_profiling_stride(self._h, every))
        self._chk(self._L.vrt_set_profiling(self._h, max_launches))

    def profile_read(self, cap=4096):
        ms = (C.c_float * cap)()
        n = self._L.vrt_profile_read(self._h, ms, cap)
        if n < 0:
            self._chk(n)
        return np.array(ms[:n], np.float32)

    def denoise(self, rgba, id_dist):
        """quad.frag's ID-aware blur through host arrays -> rgba8[H,W,4]."""
        rgba = np.ascontiguousarray(rgba, np.uint8)
        idd = np.ascontiguousarray(id_dist, np.int32)
        h, w = rgba.shape[:2]
        out = np.zeros_like(rgba)
        self._chk(self._L.vrt_denoise_host(self._h, w, h, rgba.ctypes.data, idd.ctypes.data, out.ctypes.data))
        return out

    def patch_voxel(self, world, x, y, z, max_depth=15):
        """After `world` has been edited at voxel (x, y, z): replaces the smallest enclosing sub-tree on the device
        (vrt_patch_plan / vrth_world_path_records / vrt_patch_apply). Returns the depth of the node replaced, or None
        when the edit needs a full upload (nothing was changed on the device then)."""
        H = host_lib()
        H.vrth_world_node_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
        H.vrth_world_path_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        plan = Patch()
        while max_depth >= 1:
            if self._L.vrt_patch_plan(self._h, x, y, z, max_depth, C.byref(plan)) != 0:
                return None
            if H.vrth_world_node_state(world._h, plan.path, plan.depth) == 2:
                p, n = C.c_void_p(), C.c_size_t(0)
                if H.vrth_world_path_records(world._h, plan.path, plan.depth, x, y, z, C.byref(p), C.byref(n)) != 0:
                    return None
                try:
                    self._chk(self._L.vrt_patch_apply(self._h, C.byref(plan), p, n.value))
                finally:
                    H.vrth_free(p)
                return plan.depth
            max_depth = plan.depth - 1
        return None

    def patch_box(self, world, lo, hi, max_depth=15):
        """After `world` has been edited anywhere inside the box of voxels [lo, hi] (inclusive): ONE patch for the whole box
        (vrt_patch_plan_box / vrth_world_box_records / vrt_patch_apply). Returns the depth of the node replaced, or None when the
        edit needs a full upload."""
        H = host_lib()
        H.vrth_world_node_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
        H.vrth_world_box_records.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        self._L.vrt_patch_plan_box.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(Patch)]
        lo3, hi3 = (C.c_int32 * 3)(*[int(v) for v in lo]), (C.c_int32 * 3)(*[int(v) for v in hi])
        plan = Patch()
        while max_depth >= 1:
            if self._L.vrt_patch_plan_box(self._h, lo3, hi3, max_depth, C.byref(plan)) != 0:
                return None
            if H.vrth_world_node_state(world._h, plan.path, plan.depth) == 2:
                p, n = C.c_void_p(), C.c_size_t(0)
                if H.vrth_world_box_records(world._h, plan.path, plan.depth, lo3, hi3, C.byref(p), C.byref(n)) != 0:
                    return None
                try:
                    self._chk(self._L.vrt_patch_apply(self._h, C.byref(plan), p, n.value))
                finally:
                    H.vrth_free(p)
                return plan.depth
            max_depth = plan.depth - 1
        return None

    def patch_begin(self):
        self._chk(self._L.vrt_patch_begin(self._h))

    def patch_end(self):
        self._chk(self._L.vrt_patch_end(self._h))

    def compact(self):
        self._chk(self._L.vrt_compact(self._h))

    def dispatch_views(self, width, height, tile_rows, shard, n_shards, mode, views, stream=None):
        """views: sequence of (inv_proj, inv_view, cam_pos, d_rgba8, d_id_dist) or a prepared (View * n) array;
        one launch renders them all (vrt_dispatch_views)."""
        if not isinstance(views, C.Array):
            views = make_views(views)
        self._chk(self._L.vrt_dispatch_views(self._h, width, height, tile_rows, shard, n_shards, mode, views, len(views),
                                             stream))

    def dispatch_frame(self, width, height, mode=MODE_FULL):
        """Dispatch + display pass with the intermediates kept on the device -> (shown, rgba8, id_dist) host arrays."""
        shown = np.zeros((height, width, 4), np.uint8)
        rgba = np.zeros((height, width, 4), np.uint8)
        idd = np.zeros((height, width, 2), np.int32)
        self._chk(self._L.vrt_dispatch_frame(self._h, width, height, mode, shown.ctypes.data, rgba.ctypes.data,
                                             idd.ctypes.data))
        return shown, rgba, idd

    def denoise_device(self, width, height, d_rgba, d_id, d_out, stream=None):
        self._chk(self._L.vrt_denoise(self._h, width, height, d_rgba, d_id, d_out, stream))

    def set_tile_scheduling(self, period):
        """Feedback scheduling of the tracing kernel's tiles (vrt_set_tile_scheduling): 0 = off, default 16."""
        self._L.vrt_set_tile_scheduling.argtypes = [C.c_void_p, C.c_int]
        self._chk(self._L.vrt_set_tile_scheduling(self._h, period))

    def sched_order(self, stream=None, cap=1 << 20):
        """The scheduler's current workgroup order for the shape last launched on `stream` (vrt_get_tile_order);
        empty until an order has been derived."""
        buf = np.zeros(cap, np.uint32)
        n = self._L.vrt_get_tile_order(self._h, stream, buf.ctypes.data, cap)
        if n < 0:
            self._chk(n)
        return buf[:min(n, cap)].copy()

    def sched_split_count(self, stream=None, cap=1 << 20):
        """How many groups at the head of the current order a VRT_MODE_FULL launch traces as part-tile waves (OPT_HEAVY_TILES)."""
        buf = np.zeros(cap, np.uint32)
        n = self._L.vrt_get_tile_order(self._h, stream, buf.ctypes.data, cap)
        if n < 0:
            self._chk(n)
        return int(buf[n]) if 0 < n < cap else 0

    def set_option(self, option, value):
        self._chk(self._L.vrt_set_option(self._h, option, value))

    def set_tile_order(self, enable, d_group_order=None, d_tile_cost=None):
        """caller-owned device buffers for the default kernel's group order / per-tile ticks (vrt_set_tile_order)"""
        self._chk(self._L.vrt_set_tile_order(self._h, 1 if enable else 0, d_group_order, d_tile_cost))

    def set_full_split(self, on):
        """`make AB=1` builds: the full path tracer as two kernels with cross-wave repacking (off by default)"""
        if not hasattr(self._L, "vrt_ab_set_full_split"):
            if on:
                raise VrtError("the two-kernel full path tracer exists in A/B builds only (make AB=1)")
            return
        self._chk(self._L.vrt_ab_set_full_split(self._h, 1 if on else 0))

    def set_bounce(self, refill_below, waves_per_simd):
        self._chk(self._L.vrt_ab_set_bounce(self._h, refill_below, waves_per_simd))

    def set_denoise_variant(self, v):
        """VRT_OPT_DISPLAY_KERNEL: 0 = two pixels per lane, each wave the cheaper walk (default); 2 / 3 = one walk forced; 1 = one pixel per lane (A/B builds only)."""
        self.set_option(OPT_DISPLAY_KERNEL, v)

    def synchronize(self):
        self._chk(self._L.vrt_synchronize(self._h))

    @property
    def stream(self):
        return self._L.vrt_stream(self._h)

    def set_root0_only(self, on):
        """A/B: False = rays that leave wide root 0 always walk the records of the octants around it"""
        self.set_option(OPT_EMPTY_OCTANTS, int(on))   # True/1: on, 2: on without the tighter root, False/0: off

    def set_ray_tables(self, on):
        """A/B: False = every launch runs the shader's own ray-generation prologue (no per-projection tables)"""
        self.set_option(OPT_RAY_TABLES, 1 if on else 0)

    def debug_math(self, op, x, y):
        """device probe of the kernels' arithmetic (test support library, vrt_test_math) on this context's device"""
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        out = np.zeros_like(x)
        r = test_lib().vrt_test_math(self._L.vrt_device(self._h), op, x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
        if r != 0:
            raise VrtError(f"vrt_test_math failed ({r})")
        return out


DELIVER_PEER_STORE, DELIVER_GATHER = 0, 1


class Multi:
    """vrt_multi: one context per device in this process, frames assembled on the first device (include/vrt.h)."""

    def __init__(self, devices):
        L = hip_lib()
        self._L = L
        ids = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        r = L.vrt_create_multi(len(devices), ids, C.byref(h))
        if r != 0:
            raise VrtError(f"vrt_create_multi failed ({r}): {L.vrt_multi_last_error(None).decode()}")
        self._h = h
        self.n = len(devices)

    def close(self):
        if self._h:
            self._L.vrt_destroy_multi(self._h)
            self._h = None

    def _chk(self, r):
        if r != 0:
            raise VrtError(f"vrt_multi error {r}: {self._L.vrt_multi_last_error(self._h).decode()}")

    def upload_octree(self, texels, tex_dim):
        t = np.ascontiguousarray(texels, dtype=np.uint8)
        self._chk(self._L.vrt_multi_upload_octree(self._h, t.ctypes.data if t.size else None, t.size, tex_dim))

    def set_camera(self, inv_proj, inv_view, cam_pos):
        ip, iv, cp = (np.ascontiguousarray(x, dtype=np.float32) for x in (inv_proj, inv_view, cam_pos))
        self._chk(self._L.vrt_multi_set_camera(self._h, _fptr(ip), _fptr(iv), _fptr(cp)))

    def frame_alloc(self, width, height):
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self._L.vrt_multi_frame_alloc(self._h, width, height, C.byref(a), C.byref(b)))
        return a.value, b.value

    def frame_free(self, d_rgba, d_id):
        self._chk(self._L.vrt_multi_frame_free(self._h, d_rgba, d_id))

    def dispatch(self, width, height, tile_rows, mode, delivery, d_rgba, d_id):
        self._chk(self._L.vrt_multi_dispatch(self._h, width, height, tile_rows, mode, delivery, d_rgba, d_id))

    def dispatch_frame(self, width, height, mode, d_shown):
        """vrt_multi_dispatch_frame: the displayed frame (trace + display pass in row bands with a halo) into d_shown on device 0"""
        self._L.vrt_multi_dispatch_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        r = self._L.vrt_multi_dispatch_frame(self._h, width, height, mode, C.c_void_p(d_shown))
        if r != 0:
            raise VrtError(f"vrt_multi_dispatch_frame failed ({r}): {self._L.vrt_multi_last_error(self._h).decode()}")

    def synchronize(self):
        self._chk(self._L.vrt_multi_synchronize(self._h))

    def stream(self):
        return self._L.vrt_multi_stream(self._h)

    def context(self, i):
        """the i-th device's context as a (borrowed) Context: do not close() it"""
        c = Context.__new__(Context)
        c._L = self._L
        c._h = C.c_void_p(self._L.vrt_multi_context(self._h, i))
        c.borrowed = True
        return c


def make_views(views):
    """(inv_proj, inv_view, cam_pos, d_rgba8, d_id_dist) tuples -> ctypes array of vrt_view"""
    arr = (View * len(views))()
    for v, (ip, iv, cp, p_rgba, p_id) in zip(arr, views):
        v.inv_projection[:] = [float(x) for x in ip]
        v.inv_view[:] = [float(x) for x in iv]
        v.camera_pos[:] = [float(x) for x in cp]
        v.d_rgba8, v.d_id_dist = p_rgba, p_id
    return arr


def shard_rows(height, tile_rows, shard, n_shards):
    return hip_lib().vrt_shard_rows(height, tile_rows, shard, n_shards)


def shard_row_indices(height, tile_rows, shard, n_shards):
    """Frame rows owned by a shard under the interleaved row-tile scheme, in compact order."""
    rows = []
    tiles = (height + tile_rows - 1) // tile_rows
    for t in range(shard, tiles, n_shards):
        rows.extend(range(t * tile_rows, min(height, (t + 1) * tile_rows)))
    return rows
