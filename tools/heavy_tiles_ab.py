"""VRT_OPT_HEAVY_TILES on / off: the translucent room (and a scene it must not touch) under the full path tracer with the feedback
scheduler running -- median kernel time of 64 launches, every frame's pixels against the oracle's golden hash."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest

frames = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]
w = conftest.room_world(V)
tex, dim = w.flatten()
CASES = [(k, None) for k in ("room_inside_1080p_full/mode2", "room_inside_720p_full/mode2", "room_outside_1080p_full/mode2", "room_outside_720p_full/mode2")]
CASES += [("room inside 4K (no golden: on == off)", (3840, 2160, [14.5, 30.5, 16.5, 32.0, -10.0])), ("room outside 4K (no golden: on == off)", (3840, 2160, [98.5, 34.5, 52.5, 197.0, -8.0])),
          ("room inside 540p (no golden: on == off)", (960, 540, [14.5, 30.5, 16.5, 32.0, -10.0]))]
for key, shape in CASES:
    if shape is None:
        g = frames[key]
        W, H, p = g["width"], g["height"], g["pose"]
    else:
        W, H, p = shape
        g = {}
    for on in (0, 1, 0, 1):
        ctx = V.Context(0)
        ctx.upload_octree(tex, dim)
        ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
        ctx.set_camera(ip, iv, cp)
        ctx.set_option(V.OPT_HEAVY_TILES, on)
        d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
        ok = True
        times = []
        for rep in range(8):
            ms = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 8)
            times += list(ms)
            px = ctx.device_read(d_rgba, (H, W, 4), np.uint8); idd = ctx.device_read(d_id, (H, W, 2), np.int32)
            if not g: g = {"rgba_fnv1a64": "%016x" % V.fnv1a64(px), "id_dist_fnv1a64": "%016x" % V.fnv1a64(idd)}   # the first setting's frame
            ok = ok and "%016x" % V.fnv1a64(px) == g["rgba_fnv1a64"] and "%016x" % V.fnv1a64(idd) == g["id_dist_fnv1a64"]
        t = np.array(times[16:])
        print("%-32s heavy tiles %d: median %.4f ms  min %.4f  max %.4f   pixels equal the oracle's: %s" % (key, on, np.median(t), t.min(), t.max(), ok), flush=True)
        assert ok
        ctx.close()
