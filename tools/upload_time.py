"""Times the edit path: flatten (host) -> vrt_upload_octree -> first dispatch (lazy wide-layout build) -> steady dispatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vrt_import  # noqa: E402

V = vrt_import.vrt()


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = V.Context(0)
    for name in sys.argv[1:] or ["dragon", "monu9", "nature", "terrain"]:
        w = V.World()
        t0 = time.perf_counter()
        if name == "terrain":
            w.fill_terrain(1024, 1)
        else:
            assert w.load_vox(os.path.join(root, "tests/golden/maps", name + ".vox"))
        t1 = time.perf_counter()
        tex, dim = w.flatten()
        t2 = time.perf_counter()
        ip, iv, cp, _ = V.camera_block((63.5, 60.5, 140.5), -90.0, -10.0, 640, 360)
        ctx.set_camera(ip, iv, cp)
        ctx.set_params(ctx.default_params())
        ups, firsts = [], []
        for _ in range(3):
            a = time.perf_counter()
            ctx.upload_octree(tex, dim)
            b = time.perf_counter()
            ctx.dispatch(640, 360, 0)
            c = time.perf_counter()
            ctx.dispatch(640, 360, 0)
            d = time.perf_counter()
            ups.append(b - a)
            firsts.append((c - b) - (d - c))
        print("%-8s texels %9d  build %7.1f ms  flatten %7.1f ms  upload %7.1f ms  first-dispatch extra %7.1f ms"
              % (name, len(tex) // 4, (t1 - t0) * 1e3, (t2 - t1) * 1e3, min(ups) * 1e3, min(firsts) * 1e3))


if __name__ == "__main__":
    main()
