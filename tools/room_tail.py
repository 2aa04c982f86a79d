"""How much of the translucent room's frame time is the tail of the launch (too few waves for the spread of per-tile cost)?
The same pose at growing frame sizes: the work per pixel is the same, the number of tiles per wave slot grows 4x per doubling."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest

w = conftest.room_world(V)
tex, dim = w.flatten()
ctx = V.Context(0)
ctx.upload_octree(tex, dim)
pose = {"inside": (14.5, 30.5, 16.5, 32.0, -10.0), "outside": (98.5, 34.5, 52.5, 197.0, -8.0)}
for name, p in pose.items():
    for (W, H) in ((960, 540), (1920, 1080), (3840, 2160), (7680, 4320)):
        ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
        ctx.set_camera(ip, iv, cp)
        d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
        for period in (0, 16):
            ctx.set_tile_scheduling(period)
            ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 40)
            ms = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 20)
            print("%-8s %5dx%-5d sched=%-2d  %.4f ms  %.4f ms per Mpixel" % (name, W, H, period, float(np.median(ms)), float(np.median(ms)) / (W * H / 1e6)), flush=True)
        ctx.device_free(d_rgba); ctx.device_free(d_id)
