"""Times the edit path: flatten (host) -> vrt_upload_octree -> first dispatch (lazy wide-layout build) -> steady dispatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import vrt_import  # noqa: E402

V = vrt_import.vrt()


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = V.Context(0)
    for name in sys.argv[1:] or ["dragon", "monu9", "nature", "terrain"]:
        t0 = time.perf_counter()
        w = bench.load_world(V, name)
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
        # the extension route: records straight from the pointer octree, no texel stream
        recs_t, ups2 = [], []
        for _ in range(3):
            a = time.perf_counter()
            rec, rdim = w.records()
            b = time.perf_counter()
            ctx.upload_records(rec, rdim)
            c = time.perf_counter()
            recs_t.append(b - a)
            ups2.append(c - b)
        print("%-8s records %8d  tree->records %7.1f ms  upload_records %7.1f ms" % (name, len(rec), min(recs_t) * 1e3, min(ups2) * 1e3))
        # single-voxel edits patched in place (vrt_patch_plan / vrt_patch_apply) against the reference's route
        ctx.upload_octree(tex, dim)
        ctx.dispatch(640, 360, 0)
        import numpy as np
        rng = np.random.default_rng(3)
        ts, depths, full_ts = [], [], []
        for i in range(40):
            x, y, z = (int(v) for v in rng.integers(2, 100, size=3))
            w.insert(x, y, z, 0xffd2d2ff, 3.0, 1.0, 0.0)
            a = time.perf_counter()
            d = ctx.patch_voxel(w, x, y, z)
            b = time.perf_counter()
            if d is None:
                continue
            ts.append(b - a)
            depths.append(d)
        a = time.perf_counter()
        t2, d2 = w.flatten()
        ctx.upload_octree(t2, d2)
        ctx.dispatch(640, 360, 0)
        b = time.perf_counter()
        print("%-8s %d edits patched: median %7.3f ms each (sub-tree depth %s); flatten + upload + layout %7.1f ms"
              % (name, len(ts), float(np.median(ts)) * 1e3 if ts else -1.0, sorted(set(depths)), (b - a) * 1e3))


if __name__ == "__main__":
    main()
