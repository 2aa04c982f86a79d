"""Per-tile ticks (100 MHz) of the translucent room under the full path tracer: how long is the longest wave against the frame?"""
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
import itertools
for p, (W, H) in itertools.product(((14.5, 30.5, 16.5, 32.0, -10.0), (98.5, 34.5, 52.5, 197.0, -8.0)), ((960, 540), (1920, 1080), (3840, 2160))):
    ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
    ctx.set_camera(ip, iv, cp)
    d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
    tx, ty = (W + 7) // 8, (H + 7) // 8
    n_tiles = tx * ty
    d_cost = ctx.device_alloc((n_tiles + 4) * 4)
    ctx.set_tile_scheduling(0)
    ms0 = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 10)
    ctx.set_tile_order(True, None, d_cost)
    ms = ctx.dispatch_timed(W, H, 0, H, V.MODE_FULL, d_rgba, d_id, 3)
    c = ctx.device_read(d_cost, (n_tiles,), np.uint32).astype(np.int64)
    ctx.set_tile_order(False)
    print("%dx%d: frame %.4f ms (measuring launch %.4f); tile ticks x 10 ns: max %.4f ms, p99 %.4f, median %.4f, mean %.4f; sum over 5120 slots %.4f ms" %
          (W, H, float(np.median(ms0)), float(np.median(ms)), c.max() * 1e-5, np.percentile(c, 99) * 1e-5, np.median(c) * 1e-5, c.mean() * 1e-5, c.sum() * 1e-5 / 5120))
    g = np.sort(c.reshape(-1)[:n_tiles // 4 * 4].reshape(-1, 4).max(1))[::-1]
    print("  pose", p[:3], "groups above 3/4 of the heaviest: %d, above 1/2: %d, above 1/4: %d of %d" % ((g > 0.75 * g[0]).sum(), (g > 0.5 * g[0]).sum(), (g > 0.25 * g[0]).sum(), len(g)))
    top = np.argsort(-c)[:12]
    print("  heaviest tiles (x, y, ms):", [(int(t % tx), int(t // tx), round(c[t] * 1e-5, 3)) for t in top])
    ctx.device_free(d_rgba); ctx.device_free(d_id); ctx.device_free(d_cost)
# the heaviest tile row alone (240 waves on 1024 SIMDs: every wave has its SIMD to itself) -- the critical path of the frame
W, H = 1920, 1080
p = (14.5, 30.5, 16.5, 32.0, -10.0)
ip, iv, cp, _ = V.camera_block(p[:3], p[3], p[4], W, H)
ctx.set_camera(ip, iv, cp)
d_rgba = ctx.device_alloc(W * H * 4); d_id = ctx.device_alloc(W * H * 8)
ctx.set_tile_scheduling(0)
for r0 in (728, 664, 400):
    ms = ctx.dispatch_timed(W, H, r0, r0 + 8, V.MODE_FULL, d_rgba, d_id, 10)
    print("rows %d..%d alone: %.4f ms" % (r0, r0 + 8, float(np.median(ms))))
