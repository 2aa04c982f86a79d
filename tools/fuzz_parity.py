"""One-off parity fuzz on the GPU box (not part of the test suite): random poses in and around the shipped maps,
all three modes and the display pass, HIP path vs the CPU oracle. usage: tools/fuzz_parity.py [n_poses] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
