"""Is a frame bound by its longest wave? For the bench's configurations: the kernel's time, the heaviest 8x8 tile's own ticks and all
tiles' ticks shared out over the wave slots the kernel's occupancy gives the chip (shader clock taken as ticks of the heaviest tile
cannot exceed the launch)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest

frames = json.load(open(os.path.join(ROOT, "tests/golden/frames.json")))["frames"]
CASES = [("dragon_1080p/mode0", 7), ("dragon_1080p/mode1", 7), ("dragon_1080p_full/mode2", 6), ("monu9_720p/mode0", 7), ("monu9_720p/mode1", 7),
         ("nature_4k/mode1", 7), ("terrain_1080p/mode0", 7), ("room_inside_1080p_full/mode2", 5), ("room_outside_1080p_full/mode2", 5)]
worlds = {}
for key, wpe in CASES:
    g = frames[key]
    name = g["map"]
    if name not in worlds:
        if name == "room": w = conftest.room_world(V)
        elif name == "terrain": w = conftest.terrain_world(V)
        else:
            w = V.World(); assert w.load_vox(os.path.join(ROOT, "tests/golden/maps/%s.vox" % name))
        worlds[name] = w.flatten()
    tex, dim = worlds[name]
    ctx = V.Context(0)
    ctx.upload_octree(tex, dim)
    W, H, p, mode = g["width"], g["height"], g["pose"], g["mode"]
    ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
    ctx.set_camera(ip, iv, cp)
    d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
    n_tiles = ((W + 7) // 8) * ((H + 7) // 8)
    d_cost = ctx.device_alloc((n_tiles + 4) * 4)
    ctx.dispatch_timed(W, H, 0, H, mode, d_rgba, d_id, 64)
    ms = float(np.median(ctx.dispatch_timed(W, H, 0, H, mode, d_rgba, d_id, 32)))
    ctx.set_tile_order(True, None, d_cost)
    ctx.dispatch_timed(W, H, 0, H, mode, d_rgba, d_id, 2)
    c = ctx.device_read(d_cost, (n_tiles,), np.uint32).astype(np.int64)
    ctx.set_tile_order(False)
    ghz = 2.4
    slots = 256 * 4 * wpe
    print("%-30s frame %.4f ms (scheduled)   heaviest tile %.4f ms  p99 %.4f  median %.4f   all tiles / %d slots %.4f ms   waves per slot %.1f" %
          (key, ms, c.max() / ghz * 1e-6, np.percentile(c, 99) / ghz * 1e-6, np.median(c) / ghz * 1e-6, slots, c.sum() / slots / ghz * 1e-6, n_tiles / slots), flush=True)
    ctx.close()
