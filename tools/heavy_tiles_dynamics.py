"""VRT_OPT_HEAVY_TILES over 30 scheduler periods of 16 launches each: the split count behind the order after each period and the period's median / slowest
launch in ms (the slowest is the measuring launch). Shows whether the count settles (profiles/r03_heavy_tiles_ab.txt)."""
import json, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest
frames = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]
w = conftest.room_world(V); tex, dim = w.flatten()
for key in ("room_inside_1080p_full/mode2", "room_outside_1080p_full/mode2", "room_inside_720p_full/mode2"):
    g = frames[key]; W, H, p = g["width"], g["height"], g["pose"]
    ctx = V.Context(0); ctx.upload_octree(tex, dim)
    ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H); ctx.set_camera(ip, iv, cp)
    d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
    out = []
    for rep in range(30):
        t = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 16)
        out.append("%d:%.2f/%.2f" % (ctx.sched_split_count(), float(np.median(t)), float(t.max())))
    print(key, " ".join(out), flush=True)
    ctx.close()
