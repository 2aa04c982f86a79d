"""Parity fuzz over SYNTHETIC worlds (not part of the test suite; the suite's bounded version is
test_seeded_random_scenes_uniforms_and_cameras): slabs and clusters of random materials -- opaque, emissive, glass of three
refractions, alpha-0 leaves -- placed either around the origin (content in several octants of the world: the general walk)
or in the positive octant only (the dispatcher's "nothing outside wide root 0" shortcut and its per-launch root), random
uniforms, eyes outside, beside and INSIDE the material, all three modes against the oracle.
usage: tests/fuzz/fuzz_worlds.py [n_worlds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import vrt_import  # noqa: E402

V = vrt_import.vrt()
PALETTE = [(0xa0a0a0ff, 3.0, 0.0, 0.0), (0x50b43cff, 3.0, 0.0, 0.0), (0xffd2d2ff, 3.0, 1.0, 0.0), (0x3c64dc96, 1.33, 0.0, 0.02),
           (0xc8dcff50, 1.5, 0.0, 0.0), (0xff3030ff, 3.0, 0.25, 0.0), (0x20202000, 1.2, 0.0, 0.0), (0x80ff80c0, 1.0, 0.0, 0.0)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = V.Context(0)
    bad = frames = hits = 0
    t0 = time.time()
    opaque_only = [0, 1, 2, 5, 6]    # materials a world may hold and still run the stack-free full path tracer (alpha 255 or 0)
    for case in range(n):
        w = V.World()
        # every other world holds no translucent voxel: VRT_MODE_FULL then takes the kernel without a ray stack (VRT_OPT_FULL_OPAQUE)
        # unless the eye sits inside a voxel
        palette = [PALETTE[i] for i in opaque_only] if case % 2 == 0 else PALETTE
        span = int(rng.choice([12, 40, 200]))
        positive = rng.random() < 0.6            # content in the octant [0, 1024)^3 only
        lo = 0 if positive else -span // 4
        base = rng.integers(0, 3, size=3) * int(rng.choice([0, 64, 256])) if positive else np.zeros(3, int)
        pts = []
        for _ in range(int(rng.integers(1, 5))):
            y = int(rng.integers(0, span // 2 + 1))
            x0, z0 = (int(v) for v in rng.integers(lo, span // 2, size=2))
            sx, sz = (int(v) for v in rng.integers(2, 14, size=2))
            m = palette[int(rng.integers(0, len(palette)))]
            xs, zs = np.meshgrid(np.arange(x0, x0 + sx), np.arange(z0, z0 + sz))
            xyz = np.stack([xs.ravel(), np.full(xs.size, y), zs.ravel()], axis=1) + base
            w.insert_many(xyz.astype(np.int32), np.full(len(xyz), m[0], np.uint32), m[1], m[2], m[3])
            pts.append(xyz)
        for _ in range(int(rng.integers(3, 9))):
            c = rng.integers(lo, span, size=3)
            k = int(rng.integers(1, 60))
            xyz = (c + rng.integers(-3, 4, size=(k, 3))).astype(np.int64)
            if positive:
                xyz = np.abs(xyz)
            xyz = xyz + base
            m = palette[int(rng.integers(0, len(palette)))]
            w.insert_many(xyz.astype(np.int32), np.full(k, m[0], np.uint32), m[1], m[2], m[3])
            pts.append(xyz)
        pts = np.concatenate(pts)
        tex, dim = w.flatten()
        ctx.upload_octree(tex, dim)
        for _ in range(3):
            W, H = int(rng.integers(9, 90)), int(rng.integers(7, 60))
            target = pts[int(rng.integers(0, len(pts)))] + 0.5
            kind = rng.integers(0, 4)
            if kind == 0:                        # INSIDE a voxel of the world (a solid, glass, an emitter ...)
                pos = target + rng.uniform(-0.4, 0.4, size=3)
            else:
                away = rng.normal(size=3)
                away[1] = abs(away[1]) + 0.2
                pos = target + away / np.linalg.norm(away) * rng.choice([1.7, 6.0, span * 0.5, span * 1.5, 700.0])
            look = pts[int(rng.integers(0, len(pts)))] + 0.5 if kind == 0 else target
            d = look - pos
            if not np.any(d):
                d = np.array([1.0, 0.0, 0.0])
            yaw = float(np.degrees(np.arctan2(d[2], d[0])))
            pitch = float(np.clip(np.degrees(np.arctan2(d[1], np.hypot(d[0], d[2]))), -89.0, 89.0))
            if rng.random() < 0.2:
                yaw, pitch = float(rng.choice([-180.0, -90.0, 0.0, 90.0, 45.0])), float(rng.choice([-89.0, 0.0, 89.0, -45.0]))
            cam = V.camera_block(tuple(float(v) for v in pos), yaw, pitch, W, H)[:3]
            ctx.set_camera(*cam)
            p = ctx.default_params()
            s = O.make_scene(tex, dim, *cam)
            if rng.random() < 0.5:
                gl = rng.uniform(0.0, 1.5, size=4).astype(np.float32)
                p.global_light[:] = [float(v) for v in gl]
                s.global_light[:] = [float(v) for v in gl]
            if rng.random() < 0.5:
                ld = rng.normal(size=3)
                ld = (ld / np.linalg.norm(ld)).astype(np.float32)
                if rng.random() < 0.3:
                    ld[int(rng.integers(0, 3))] = 0.0
                p.light_dir[:] = [float(v) for v in ld]
                s.light_dir[:] = [float(v) for v in ld]
            if rng.random() < 0.3:
                hv = pts[int(rng.integers(0, len(pts)))]
                p.highlighted[:] = [int(v) for v in hv]
                s.highlighted[:] = [int(v) for v in hv]
            if rng.random() < 0.3:
                vs = float(np.float32(rng.choice([0.5, 2.0, 1.25])))
                p.voxel_scale = vs
                s.voxel_scale = vs
            ctx.set_params(p)
            for mode in (0, 1, 2):
                ref_rgba, ref_id, _, _ = O.render(s, W, H, mode)
                rgba, idd = ctx.dispatch(W, H, mode)
                frames += 1
                hits += int(np.count_nonzero(ref_id[..., 0]))
                if not (np.array_equal(rgba, ref_rgba) and np.array_equal(idd, ref_id)):
                    bad += 1
                    print("MISMATCH world", case, "positive" if positive else "around origin", "span", span, "base", base.tolist(), "mode", mode,
                          W, H, "pos", [float(v) for v in pos], yaw, pitch, "kind", int(kind),
                          int(np.count_nonzero(np.any(rgba != ref_rgba, axis=-1) | np.any(idd != ref_id, axis=-1))), "px", flush=True)
        if case % 20 == 19:
            print(case + 1, "worlds,", frames, "frames,", bad, "mismatches, %.0f s" % (time.time() - t0), flush=True)
    ctx.set_params(ctx.default_params())
    print("world fuzz done:", n, "worlds,", frames, "frames,", hits, "hit pixels,", bad, "mismatches")
    sys.exit(1 if bad else 0)


main()
