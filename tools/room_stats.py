"""Reads one quantity per 8x8 tile of the room's 1080p frame from an experiment build (tools/room_stats.sh; VRT_HIP_LIB names it;
without it: the product library's tile ticks) and saves it as gpurun_out/room_stats_<name>.npy."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vrt_import
V = vrt_import.vrt()
import conftest

name = sys.argv[1]
w = conftest.room_world(V)
tex, dim = w.flatten()
ctx = V.Context(0)
ctx.upload_octree(tex, dim)
p = (14.5, 30.5, 16.5, 32.0, -10.0)
W, H = 1920, 1080
ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
ctx.set_camera(ip, iv, cp)
d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
n_tiles = (W // 8) * (H // 8)
d_cost = ctx.device_alloc((n_tiles + 4) * 4)
ctx.set_tile_scheduling(0)
ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 3)
ctx.set_tile_order(True, None, d_cost)
ms = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 2)
c = ctx.device_read(d_cost, (n_tiles,), np.uint32).astype(np.int64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.save(os.path.join(ROOT, "gpurun_out", "room_stats_%s.npy" % name), c)
print(name, "frame ms", np.round(ms, 4), "max", c.max(), "mean", round(float(c.mean()), 1), "tile (53,91):", c[91 * 240 + 53])
