"""One frame of a fuzz case against the oracle, pixel by pixel: tests/fuzz/repro_case.py map W H x y z yaw pitch [mode]
Prints the differing pixels for the default variant with and without the ray-generation tables and for variant 20 (v3)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import vrt_import  # noqa: E402

V = vrt_import.vrt()


def main():
    name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    pos = tuple(float(v) for v in sys.argv[4:7])
    yaw, pitch = float(sys.argv[7]), float(sys.argv[8])
    mode = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    w = V.World()
    assert w.load_vox(os.path.join(ROOT, "tests/golden/maps", name + ".vox"))
    tex, dim = w.flatten()
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    cam = V.camera_block(pos, yaw, pitch, W, H)[:3]
    ctx.set_camera(*cam)
    s = O.make_scene(tex, dim, *cam)
    ref_rgba, ref_id, fm, st = O.render(s, W, H, mode, want_fetch_map=True)
    print("oracle hits", int(np.count_nonzero(ref_id[..., 0])), "of", W * H, st)
    for label, variant, tables in (("v4 tables", 0, True), ("v4 no tables", 0, False), ("v3", 20, True)):
        ctx.set_variant(variant)
        ctx.set_ray_tables(tables)
        rgba, idd = ctx.dispatch(W, H, mode)
        bad = np.argwhere(np.any(rgba != ref_rgba, axis=-1) | np.any(idd != ref_id, axis=-1))
        print(label, ":", len(bad), "pixels differ")
        for (y, x) in bad[:12]:
            print("   (x=%d, y=%d) got rgba %s id %s   want rgba %s id %s   oracle fetches %d" %
                  (x, y, rgba[y, x].tolist(), idd[y, x].tolist(), ref_rgba[y, x].tolist(), ref_id[y, x].tolist(), int(fm[y, x])))


main()
