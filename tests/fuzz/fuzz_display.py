"""Parity fuzz of the display pass on frames large enough to fill its 32 x 16 tiles: random poses of the shipped maps (and the room) under
the full path tracer at mid sizes, the pass in its three settings (each wave the cheaper walk / common rows / own boxes) against the
oracle's quad.frag restatement; widths that are and are not multiples of four (16-byte and tap-by-tap staging).
Usage (GPU box): python tests/fuzz/fuzz_display.py <seed> <poses per map>"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_py as O
import vrt_import
V = vrt_import.vrt()
import conftest

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
t0 = time.time()
frames = bad = 0
for name in ("dragon", "monu9", "nature", "room"):
    if name == "room":
        w = conftest.room_world(V); lo, hi = (-30, 5, -30), (100, 70, 100)
    else:
        w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps/%s.vox" % name)); lo, hi = (-40, 5, -40), (170, 150, 260)
    ctx = V.Context(0)
    ctx.upload_octree(*w.flatten())
    for i in range(n):
        W, H = [(640, 360), (514, 290), (400, 300), (333, 207)][i % 4]
        pos = rng.uniform(lo, hi)
        # look at a point inside the model (Camera.hpp: front = (cos yaw cos pitch, sin pitch, sin yaw cos pitch)), now and then anywhere
        tgt = rng.uniform((20, 10, 20), (110, 90, 110)) if name != "room" else rng.uniform((5, 20, 5), (45, 45, 45))
        d = tgt - pos
        yaw, pitch = np.degrees(np.arctan2(d[2], d[0])), float(np.clip(np.degrees(np.arcsin(d[1] / max(np.linalg.norm(d), 1e-6))), -80, 80))
        if i % 7 == 6: yaw, pitch = rng.uniform(-180, 180), rng.uniform(-60, 30)
        ip, iv, cp, _ = V.camera_block(tuple(float(x) for x in pos), float(yaw), float(pitch), W, H)
        ctx.set_camera(ip, iv, cp)
        rgba, idd = ctx.dispatch(W, H, 2)
        if not np.any(idd[..., 0]):
            continue
        ref = O.denoise(rgba, idd)
        for dv in (0, 2, 3):
            ctx.set_denoise_variant(dv)
            got = ctx.denoise(rgba, idd)
            frames += 1
            if not np.array_equal(got, ref):
                bad += 1
                print("MISMATCH", name, i, W, H, "kernel", dv, "pos", pos.tolist(), yaw, pitch, int(np.count_nonzero(np.any(got != ref, axis=2))), "pixels", flush=True)
        ctx.set_denoise_variant(0)
    print("%s: %d poses, %d display frames so far, %d mismatches, %.0f s" % (name, n, frames, bad, time.time() - t0), flush=True)
    ctx.close()
print("display fuzz done: seed %d, %d frames, %d mismatches" % (seed, frames, bad))
sys.exit(1 if bad else 0)
