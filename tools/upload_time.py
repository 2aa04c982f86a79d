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
        # box edits: a fill of n^3 voxels as ONE patch (vrt_patch_plan_box / vrth_world_box_records) against n^3 single-voxel
        # patches in a batch and against the reference's route (re-flatten + upload on every click, src/main.cpp:843-914)
        for n in (4, 16):
            one, many = [], []
            for trial in range(6):
                lo = [int(v) for v in rng.integers(2, 90, size=3)]
                if trial % 2 == 0:
                    lo = [v & ~(n - 1) for v in lo]          # aligned: the box is one octree node's worth
                hi = [v + n - 1 for v in lo]
                g = np.indices((n, n, n)).reshape(3, -1).T + np.array(lo)
                col = np.full(len(g), 0x3296c8ff if trial & 1 else 0xc86432ff, np.uint32)
                w.insert_many(g.astype(np.int32), col)
                a = time.perf_counter()
                d = ctx.patch_box(w, lo, hi)
                b = time.perf_counter()
                if d is not None:
                    one.append((b - a, d))
                # the same fill elsewhere as n^3 voxel patches in one batch
                lo2 = [v + 100 for v in lo]
                g2 = g + 100
                a = time.perf_counter()
                ctx.patch_begin()
                ok = True
                for (x, y, z), c_ in zip(g2, col):   # a voxel patch describes ONE edit: insert, patch, insert, patch ...
                    w.insert(int(x), int(y), int(z), int(c_))
                    ok = ok and ctx.patch_voxel(w, int(x), int(y), int(z)) is not None
                    if not ok:
                        break
                ctx.patch_end()
                b = time.perf_counter()
                if ok:
                    many.append(b - a)
                else:
                    t3, d3 = w.flatten()
                    ctx.upload_octree(t3, d3)
            print("%-8s %2d^3 fill: one box patch median %7.3f ms (depths %s), %d voxel patches in a batch median %8.3f ms"
                  % (name, n, float(np.median([t for t, _ in one])) * 1e3 if one else -1.0, sorted({d for _, d in one}), n ** 3,
                     float(np.median(many)) * 1e3 if many else -1.0))


if __name__ == "__main__":
    main()
