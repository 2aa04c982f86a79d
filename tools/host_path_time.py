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


    # the asynchronous, double-buffered form into page-locked buffers: two frames in flight
    for both in (True, False):
        bufs = [(ctx.host_alloc((H, W, 4), np.uint8), ctx.host_alloc((H, W, 2), np.int32) if both else None) for _ in range(2)]
        for mode, label in ((0, "primary"), (2, "full")):
            tickets = []
            for i in range(4):
                tickets.append(ctx.dispatch_async(W, H, mode, *bufs[i & 1]))
            for t in set(tickets):
                ctx.dispatch_wait(t)
            n = 40
            t0 = time.perf_counter()
            for i in range(n):
                ctx.dispatch_async(W, H, mode, *bufs[i & 1])
            ctx.dispatch_wait(0)
            ctx.dispatch_wait(1)
            dt = (time.perf_counter() - t0) / n
            ok = np.array_equal(bufs[0][0], bufs[1][0]) and (not both or np.array_equal(bufs[0][1], bufs[1][1]))
            print("%-58s %8.3f ms per frame  (both lanes hold the same frame: %s)" %
                  (f"vrt_dispatch_async {label} -> pinned rgba" + (" + id/dist" if both else " only"), dt * 1e3, ok))
        for a, b in bufs:
            ctx.host_free(a)
            if b is not None:
                ctx.host_free(b)


if __name__ == "__main__":
    main()
