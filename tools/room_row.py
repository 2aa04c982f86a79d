"""The heaviest tile row of the translucent room alone (rows 728..736 of the 1080p inside pose, full path tracer): the frame's
critical path, launched a few times for the counters (tools/pmc_room_row.sh)."""
import os, sys
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
p = (14.5, 30.5, 16.5, 32.0, -10.0)
W, H = 1920, 1080
ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
ctx.set_camera(ip, iv, cp)
d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
ctx.set_tile_scheduling(0)
r0 = int(os.environ.get("VRT_ROW0", "728"))
ms = ctx.dispatch_timed(W, H, r0, r0 + 8, V.MODE_FULL, d_rgba, d_id, int(os.environ.get("VRT_ITERS", "5")))
print("rows %d..%d alone: %s ms" % (r0, r0 + 8, np.round(ms, 4)))
ms = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 10)
import json
g = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]["room_inside_1080p_full/mode2"]
px = ctx.device_read(d_rgba, (H, W, 4), np.uint8)
print("whole frame, unscheduled: %.4f ms; pixels equal the oracle's: %s" % (float(np.median(ms)), "%016x" % V.fnv1a64(px) == g["rgba_fnv1a64"]))
