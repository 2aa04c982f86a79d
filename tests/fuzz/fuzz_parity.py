"""One-off parity fuzz on the GPU box (not part of the test suite): random poses in and around the shipped maps,
all three modes and the display pass, HIP path vs the CPU oracle. usage: tests/fuzz/fuzz_parity.py [n_poses] [seed]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = V.Context(0)
    bad = frames = hits = pixels = 0
    t0 = time.time()
    for name, extent in (("dragon", (128, 110, 60)), ("monu9", (100, 120, 100)), ("nature", (128, 160, 128))):
        w = V.World()
        assert w.load_vox(os.path.join(ROOT, "tests/golden/maps", name + ".vox"))
        tex, dim = w.flatten()
        ctx.upload_octree(tex, dim)
        for i in range(n):
            W, H = int(rng.integers(8, 97)), int(rng.integers(8, 65))
            ext = np.array(extent, float)
            kind = rng.integers(0, 4)
            if kind == 0:      # inside the bounding box
                pos = rng.uniform(0, 1, 3) * ext
            elif kind == 1:    # around it
                pos = ext / 2 + rng.normal(size=3) * ext
            elif kind == 2:    # on integer / half-integer coordinates (voxel faces and centres)
                pos = np.round(rng.uniform(-0.2, 1.2, 3) * ext * 2) / 2
            else:              # far away, some outside the world
                pos = ext / 2 + rng.normal(size=3) * 900.0
            yaw = float(rng.choice([rng.uniform(-180, 180), rng.choice([-180.0, -90.0, 0.0, 90.0, 45.0])]))
            pitch = float(rng.choice([rng.uniform(-89, 89), rng.choice([-89.0, 0.0, 89.0, -45.0])]))
            cam = V.camera_block(tuple(float(v) for v in pos), yaw, pitch, W, H)[:3]
            ctx.set_camera(*cam)
            p = ctx.default_params()
            s = O.make_scene(tex, dim, *cam)
            if rng.random() < 0.3:
                hl = [int(v) for v in rng.integers(0, 100, size=3)]
                p.highlighted[:] = hl
                s.highlighted[:] = hl
            ctx.set_params(p)
            for mode in (0, 1, 2):
                ref_rgba, ref_id, _, _ = O.render(s, W, H, mode)
                rgba, idd = ctx.dispatch(W, H, mode)
                frames += 1
                hits += int(np.count_nonzero(ref_id[..., 0]))
                pixels += W * H
                if not (np.array_equal(rgba, ref_rgba) and np.array_equal(idd, ref_id)):
                    bad += 1
                    print("MISMATCH", name, i, "mode", mode, W, H, "pos", pos.tolist(), yaw, pitch,
                          int(np.count_nonzero(np.any(rgba != ref_rgba, axis=-1))), "px", flush=True)
                elif mode == 2 and not np.array_equal(ctx.denoise(rgba, idd), O.denoise(rgba, idd)):
                    bad += 1
                    print("MISMATCH display pass", name, i, W, H, flush=True)
            if i % 25 == 24:
                print(name, i + 1, "poses,", frames, "frames,", bad, "mismatches, %.0f s" % (time.time() - t0), flush=True)
    print("fuzz done:", frames, "frames,", pixels, "pixels,", hits, "with a hit,", bad, "mismatches")
    # second phase: the feedback tile scheduler. Frames large enough to be scheduled (>= 2048 groups of four tiles),
    # a camera that keeps moving, orders re-derived every 1-3 launches; every frame against the same frame traced
    # with the scheduler off (which the first phase ties to the oracle).
    sched_frames = sched_bad = 0
    n2 = max(10, n // 25)
    for name, extent in (("dragon", (128, 110, 60)), ("monu9", (100, 120, 100)), ("nature", (128, 160, 128))):
        w = V.World()
        assert w.load_vox(os.path.join(ROOT, "tests/golden/maps", name + ".vox"))
        ctx.upload_octree(*w.flatten())
        ext = np.array(extent, float)
        for shape in ((1024, 520), (1352, 760), (1920, 1080)):
            W, H = shape
            for mode in (0, 1, 2):
                pos = ext / 2 + rng.normal(size=3) * ext
                yaw, pitch = float(rng.uniform(-180, 180)), float(rng.uniform(-60, 60))
                period = int(rng.integers(1, 4))
                for i in range(n2):
                    pos = pos + rng.normal(size=3) * 2.0       # a walk: orders are always a little stale
                    yaw += float(rng.normal() * 3.0)
                    if rng.random() < 0.1:                      # and sometimes completely
                        pos = rng.uniform(0, 1, 3) * ext
                    ctx.set_camera(*V.camera_block(tuple(float(v) for v in pos), yaw, pitch, W, H)[:3])
                    ctx.set_tile_scheduling(0)
                    ref = ctx.dispatch(W, H, mode)
                    ref_shown = ctx.denoise(*ref) if mode == 2 else None
                    ctx.set_tile_scheduling(period)
                    got = ctx.dispatch(W, H, mode)
                    sched_frames += 1
                    if not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
                        sched_bad += 1
                        print("MISMATCH scheduled", name, shape, "mode", mode, "frame", i, flush=True)
                    o = ctx.sched_order()
                    if mode == 2 and not np.array_equal(ctx.denoise(*got), ref_shown):   # the display pass, scheduled too
                        sched_bad += 1
                        print("MISMATCH scheduled display pass", name, shape, "frame", i, flush=True)
                    if o.size == 0 and i == 0 and period > 1:
                        continue   # a shape's first launch is never the measured one: no order yet
                    if not np.array_equal(np.sort(o), np.arange(o.size, dtype=np.uint32)) or o.size == 0:
                        sched_bad += 1
                        print("BAD ORDER", name, shape, "mode", mode, "frame", i, o.size, flush=True)
        print(name, "scheduled frames", sched_frames, "mismatches", sched_bad, "%.0f s" % (time.time() - t0), flush=True)
    print("scheduler fuzz done:", sched_frames, "frames,", sched_bad, "mismatches")
    sys.exit(1 if (bad or sched_bad) else 0)


if __name__ == "__main__":
    main()
