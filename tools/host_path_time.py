"""Frame time through the HOST-buffer entry points (what a main.cpp-style caller with std::vector images sees)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrt_import  # noqa: E402

V = vrt_import.vrt()


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W, H = 1920, 1080
    w = V.World()
    assert w.load_vox(os.path.join(root, "tests/golden/maps/dragon.vox"))
    tex, dim = w.flatten()
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, W, H)
    ctx.set_camera(ip, iv, cp)
    ctx.set_params(ctx.default_params())
    L, h = ctx._L, ctx._h
    rgba = np.zeros((H, W, 4), np.uint8)
    idd = np.zeros((H, W, 2), np.int32)
    shown = np.zeros((H, W, 4), np.uint8)
    for name, fn in [
        ("vrt_dispatch primary -> rgba + id/dist", lambda: L.vrt_dispatch(h, W, H, 0, rgba.ctypes.data, idd.ctypes.data)),
        ("vrt_dispatch primary -> rgba only", lambda: L.vrt_dispatch(h, W, H, 0, rgba.ctypes.data, None)),
        ("vrt_dispatch full -> rgba + id/dist", lambda: L.vrt_dispatch(h, W, H, 2, rgba.ctypes.data, idd.ctypes.data)),
        ("vrt_denoise_host", lambda: L.vrt_denoise_host(h, W, H, rgba.ctypes.data, idd.ctypes.data, shown.ctypes.data)),
        ("vrt_dispatch_frame full -> shown only", lambda: L.vrt_dispatch_frame(h, W, H, 2, shown.ctypes.data, None, None)),
    ]:
        for _ in range(3):
            assert fn() == 0
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            fn()
        print("%-42s %8.3f ms per call" % (name, (time.perf_counter() - t0) / n * 1e3))


if __name__ == "__main__":
    main()
